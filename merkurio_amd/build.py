"""Builds libmerkurio_hip.so (hand-written gfx950 kernels + C-ABI host code) in-tree with hipcc.

    python -m merkurio_amd.build [--force]

hipcc cross-compiles for gfx950 without a GPU; the built .so travels to the GPU box with the
repository snapshot (it is git-ignored, not gpurun-ignored).
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB_DIR = os.path.join(HERE, "lib")
LIB_PATH = os.path.join(LIB_DIR, "libmerkurio_hip.so")
CLI_PATH = os.path.join(LIB_DIR, "merkurio")

LIB_SOURCES = ["scan_kernel.hip", "matcher.cpp", "host_patterns.cpp", "host_loops.cpp"]
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-result"]
# profiling builds: MERKURIO_HIPCC_FLAGS="-DMK_ABLATE=1" python -m merkurio_amd.build --force
FLAGS += os.environ.get("MERKURIO_HIPCC_FLAGS", "").split()


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def _deps():
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC)]
    deps.append(os.path.join(os.path.dirname(HERE), "include", "merkurio_hip.h"))
    return deps


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
            tmp = LIB_PATH + ".tmp.%d" % os.getpid()
            cmd = [HIPCC, *FLAGS, "-shared", "-o", tmp] + [os.path.join(CSRC, s) for s in LIB_SOURCES]
            if verbose:
                print(" ".join(cmd), flush=True)
            subprocess.run(cmd, check=True)
            os.replace(tmp, LIB_PATH)
    return LIB_PATH


def build_cli(force=False, verbose=False):
    """The C++ `merkurio extract|tag` host program (links libmerkurio_hip.so)."""
    srcs = [os.path.join(CSRC, "cli", f) for f in sorted(os.listdir(os.path.join(CSRC, "cli")))
            if f.endswith(".cpp")] if os.path.isdir(os.path.join(CSRC, "cli")) else []
    if not srcs:
        return None
    deps = srcs + [os.path.join(CSRC, "cli", f) for f in os.listdir(os.path.join(CSRC, "cli"))] + [LIB_PATH]
    if not force and not _stale(CLI_PATH, deps):
        return CLI_PATH
    cmd = [HIPCC, "-O2", "-std=c++17", "-Wall", "-Wno-unused-result", "-o", CLI_PATH, *srcs, "-L" + LIB_DIR,
           "-lmerkurio_hip", "-lz", "-lpthread", "-Wl,-rpath,$ORIGIN"]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.run(cmd, check=True)
    return CLI_PATH


def build_all(force=False, verbose=False):
    lib = build_lib(force, verbose)
    cli = build_cli(force, verbose)
    return lib, cli


if __name__ == "__main__":
    print(build_all(force="--force" in sys.argv, verbose=True))
