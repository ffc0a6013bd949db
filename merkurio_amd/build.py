"""Builds libmerkurio_hip.so (hand-written gfx950 kernels + C-ABI host code) in-tree with hipcc.

    python -m merkurio_amd.build [--force] [--tag NAME --flags "-DMK_ABLATE=1 ..."]

hipcc cross-compiles for gfx950 without a GPU; the built .so travels to the GPU box with the
repository snapshot (it is git-ignored, not gpurun-ignored).  The scan kernel has ~80 template
variants; they are compiled as independent translation units (scan_variants.hip with
-DMK_TU=n) in parallel and linked into one library.

--tag NAME builds merkurio_amd/lib/libmerkurio_hip_NAME.so with extra flags (A/B and ablation
builds for the scripts under tools/; select one at run time with MERKURIO_LIB_PATH).
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB_DIR = os.path.join(HERE, "lib")
LIB_PATH = os.path.join(LIB_DIR, "libmerkurio_hip.so")
CLI_PATH = os.path.join(LIB_DIR, "merkurio")

N_VARIANT_TUS = 17
HOST_SOURCES = ["matcher.cpp", "host_patterns.cpp", "host_loops.cpp", "reduce.cpp"]
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-result"]
# profiling builds: MERKURIO_HIPCC_FLAGS="-DMK_ABLATE=1" python -m merkurio_amd.build --force
FLAGS += os.environ.get("MERKURIO_HIPCC_FLAGS", "").split()
JOBS = int(os.environ.get("MERKURIO_BUILD_JOBS", str(min(8, os.cpu_count() or 1))))


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def _headers():
    hs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".h", ".hpp"))]
    hs += [os.path.join(CSRC, "codec", f) for f in os.listdir(os.path.join(CSRC, "codec")) if f.endswith((".h", ".hpp"))]
    hs.append(os.path.join(os.path.dirname(HERE), "include", "merkurio_hip.h"))
    return hs


def _units():
    """(object name, source file, extra flags) of every translation unit of the library"""
    units = [("scan_tu%d.o" % n, "scan_variants.hip", ["-DMK_TU=%d" % n]) for n in range(N_VARIANT_TUS)]
    units.append(("scan_kernel.o", "scan_kernel.hip", []))
    units.append(("order_hits.o", "order_hits.hip", []))
    units.append(("order_hits_fallback.o", "order_hits_fallback.hip", []))
    units.append(("sets.o", "sets.hip", []))
    units.append(("build_tables.o", "build_tables.hip", []))
    units.append(("ingest.o", "ingest.hip", []))
    units.append(("bam.o", "bam.hip", []))
    # the device BGZF codec (v5 of the ABI)
    units.append(("codec_bgzf_deflate.o", "codec/bgzf_deflate.hip", []))
    units.append(("codec_bgzf_inflate.o", "codec/bgzf_inflate.hip", []))
    units.append(("codec_bgzf_inflate_wave.o", "codec/bgzf_inflate_wave.hip", []))
    units.append(("codec_gzip_inflate.o", "codec/gzip_inflate.hip", []))
    units.append(("codec_gzip_segments_wave.o", "codec/gzip_segments_wave.hip", []))
    units.append(("codec_host.o", "codec/codec_host.cpp", []))
    units += [(s.replace(".cpp", ".o"), s, []) for s in HOST_SOURCES if os.path.exists(os.path.join(CSRC, s))]
    return units


def _deps():
    srcs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if os.path.isfile(os.path.join(CSRC, f))]
    srcs += [os.path.join(CSRC, "codec", f) for f in os.listdir(os.path.join(CSRC, "codec"))]
    return srcs + _headers()


def _parse_resource_remarks(stderr):
    """-Rpass-analysis=kernel-resource-usage -> rows (kernel, vgprs, agprs, sgprs, scratch bytes/lane, LDS
    bytes/block, waves/SIMD) and the rest of the compiler's output (real warnings and errors)"""
    import re
    rows, cur, rest = [], None, []
    for line in stderr.splitlines():
        if "[-Rpass-analysis=kernel-resource-usage]" not in line:
            rest.append(line)
            continue
        m = re.search(r"remark:\s+(.*?)\s*\[-Rpass", line)
        if not m:
            continue
        k, _, v = m.group(1).partition(":")
        k, v = k.strip(), v.strip()
        if k == "Function Name":
            cur = {"name": v}
            rows.append(cur)
        elif cur is not None:
            cur[k] = v
    out = []
    for r in rows:
        name = r["name"]
        for tool in ("c++filt", "/opt/rocm/lib/llvm/bin/llvm-cxxfilt"):  # readable names; the mangled one if neither exists
            try:
                name = subprocess.run([tool, name], capture_output=True, text=True).stdout.strip() or name
                break
            except OSError:
                continue
        out.append((name, r.get("VGPRs", "?"), r.get("AGPRs", "?"), r.get("TotalSGPRs", r.get("SGPRs", "?")),
                    r.get("ScratchSize [bytes/lane]", "?"), r.get("LDS Size [bytes/block]", "?"), r.get("Occupancy [waves/SIMD]", "?")))
    return out, "\n".join(rest) + ("\n" if rest else "")


def isa_resources(obj_dir, out_file, strict=True):
    """Collects the per-kernel resource records of every device translation unit into one table and
    REFUSES a build in which a kernel spills: the scan kernel's design rests on "no scratch" (a
    spilled variant reaches its LDS rings through flat instructions: 3x slower)."""
    rows = []
    for f in sorted(os.listdir(obj_dir)):
        if f.endswith(".isa.txt"):
            with open(os.path.join(obj_dir, f)) as fh:
                rows += [line.rstrip("\n").split("\t") for line in fh if line.strip()]
    if not rows:
        return None
    rows.sort(key=lambda r: r[0])
    with open(out_file, "w") as f:
        f.write("# kernel\tVGPRs\tAGPRs\tSGPRs\tscratch bytes/lane\tLDS bytes/block (static)\twaves/SIMD\n")
        for r in rows:
            f.write("\t".join(r) + "\n")
    # (the library sort of the fallback path, rocPRIM's merge sort, spills by itself: recorded, not refused)
    # (matched in both forms: a host without a demangler keeps the mangled name, _ZN7rocprim...)
    # (r05: the codec kernels are held to the same rule -- their block headers no longer keep arrays in private memory)
    spilled = [r[0] for r in rows if r[4] not in ("0", "?") and "rocprim" not in r[0]]
    if spilled and strict:
        raise RuntimeError("kernels with scratch memory (register spills): " + "; ".join(spilled))
    if spilled:  # A/B and ablation builds (--tag): say so, keep going
        print("note: kernels with scratch memory in this tagged build: " + "; ".join(spilled), file=sys.stderr)
    return out_file


def _compile_lib(out_path, extra_flags, obj_dir, force, verbose):
    os.makedirs(obj_dir, exist_ok=True)
    headers = _headers()
    jobs = []
    objs = []
    for obj, src, fl in _units():
        o = os.path.join(obj_dir, obj)
        s = os.path.join(CSRC, src)
        objs.append(o)
        if force or _stale(o, [s] + headers):
            jobs.append([HIPCC, *FLAGS, *extra_flags, *fl, "-c", "-o", o, s])

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        if cmd[-1].endswith(".hip") and "-c" in cmd:
            # device code: keep the compiler's per-kernel resource remarks next to the object
            # (registers, scratch, LDS, occupancy) -- see isa_resources() below
            r = subprocess.run(cmd + ["-Rpass-analysis=kernel-resource-usage"], stderr=subprocess.PIPE, text=True)
            obj = cmd[cmd.index("-o") + 1]
            rows, rest = _parse_resource_remarks(r.stderr)
            if rest.strip():
                sys.stderr.write(rest)
            if r.returncode:
                raise subprocess.CalledProcessError(r.returncode, cmd)
            with open(obj + ".isa.txt", "w") as f:
                for row in rows:
                    f.write("\t".join(str(x) for x in row) + "\n")
            return
        subprocess.run(cmd, check=True)

    if jobs:
        with ThreadPoolExecutor(max(1, JOBS)) as ex:
            list(ex.map(run, jobs))
    for f in os.listdir(obj_dir):  # objects of translation units that no longer exist
        if (f.endswith(".o") or f.endswith(".isa.txt")) and os.path.join(obj_dir, f.replace(".isa.txt", "")) not in objs:
            os.remove(os.path.join(obj_dir, f))
    isa_resources(obj_dir, os.path.join(os.path.dirname(out_path), "isa_resources" + os.path.basename(out_path)[len("libmerkurio_hip"):-3] + ".txt"),
                  strict=not extra_flags)
    if jobs or force or _stale(out_path, objs):
        tmp = out_path + ".tmp.%d" % os.getpid()
        run([HIPCC, "--offload-arch=gfx950", "-shared", "-o", tmp, *objs, "-ldl", "-lpthread"])
        os.replace(tmp, out_path)
    return out_path


def build_lib(force=False, verbose=False):
    """(Re)builds the library if it is missing or older than its sources.  Serialised with a
    file lock and written through a temporary name: several ranks of one job may call this."""
    import fcntl
    os.makedirs(LIB_DIR, exist_ok=True)
    if not force and not _stale(LIB_PATH, _deps()):
        return LIB_PATH
    with open(os.path.join(LIB_DIR, ".build.lock"), "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        if force or _stale(LIB_PATH, _deps()):  # still stale once we hold the lock
            _compile_lib(LIB_PATH, [], os.path.join(LIB_DIR, "obj"), force, verbose)
    return LIB_PATH


def build_tagged(tag, flags, force=False, verbose=False, only_scan=True):
    """A/B or ablation build: libmerkurio_hip_<tag>.so compiled with extra `flags`."""
    out = os.path.join(LIB_DIR, "libmerkurio_hip_%s.so" % tag)
    return _compile_lib(out, list(flags), os.path.join(LIB_DIR, "obj_" + tag), force, verbose)


def build_cli(force=False, verbose=False):
    """The C++ `merkurio extract|tag` host program (links libmerkurio_hip.so)."""
    srcs = [os.path.join(CSRC, "cli", f) for f in sorted(os.listdir(os.path.join(CSRC, "cli")))
            if f.endswith(".cpp")] if os.path.isdir(os.path.join(CSRC, "cli")) else []
    if not srcs:
        return None
    deps = srcs + [os.path.join(CSRC, "cli", f) for f in os.listdir(os.path.join(CSRC, "cli"))] + [LIB_PATH]
    if not force and not _stale(CLI_PATH, deps):
        return CLI_PATH
    obj_dir = os.path.join(LIB_DIR, "obj_cli")
    os.makedirs(obj_dir, exist_ok=True)
    cli_hdrs = [os.path.join(CSRC, "cli", f) for f in os.listdir(os.path.join(CSRC, "cli")) if f.endswith(".hpp")] + _headers()
    jobs, objs = [], []
    for s in srcs:
        o = os.path.join(obj_dir, os.path.basename(s).replace(".cpp", ".o"))
        objs.append(o)
        if force or _stale(o, [s] + cli_hdrs):
            jobs.append([HIPCC, "--offload-arch=gfx950", "-O2", "-std=c++17", "-Wall", "-Wno-unused-result", "-c", "-o", o, s])

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.run(cmd, check=True)

    with ThreadPoolExecutor(max(1, JOBS)) as ex:
        list(ex.map(run, jobs))
    run([HIPCC, "-o", CLI_PATH, *objs, "-L" + LIB_DIR, "-lmerkurio_hip", "-lz", "-ldl", "-lpthread", "-Wl,-rpath,$ORIGIN"])
    return CLI_PATH


def build_all(force=False, verbose=False):
    lib = build_lib(force, verbose)
    cli = build_cli(force, verbose)
    return lib, cli


if __name__ == "__main__":
    if "--tag" in sys.argv:
        tag = sys.argv[sys.argv.index("--tag") + 1]
        fl = sys.argv[sys.argv.index("--flags") + 1].split() if "--flags" in sys.argv else []
        print(build_tagged(tag, fl, force="--force" in sys.argv, verbose=True))
    else:
        print(build_all(force="--force" in sys.argv, verbose=True))
