"""ctypes binding of libmerkurio_hip.so (include/merkurio_hip.h) and a thin Python mirror of the
reference's matcher surface, so tests read like the reference's own tests:

    BNDMq(pattern, q).find_iter(text) / find_all / find_match   src/pattern_matching.rs:61-153
    AhoCorasick(patterns, ascii_case_insensitive).find_overlapping_iter(text)
                                                                 src/cmd_extract.rs:260-265,332
    parse_pattern_list / read_kmers_from_file / recommend_aho_corasick / tune_q_value /
    generate_masks                                               src/helpers.rs, pattern_*.rs
    Matcher(...).extract_single / extract_paired / tag_records   the record loops of cmd_*.rs

Everything that searches text runs in the gfx950 kernels; there is no CPU fallback here: if
the library or a GPU is missing, construction raises.
"""
import ctypes as C
import time
import importlib.util
import os

import numpy as np

from . import build as _build

MK_OK = 0
MK_E_EMPTY_PATTERN = -1
MK_E_INVALID_Q = -2
MK_E_PATTERN_TOO_LONG = -3
MK_E_NO_PATTERNS = -4
MK_E_NOMEM = -5
MK_E_PAIR_MISMATCH = -6
MK_E_HIP = -7
MK_E_CAPACITY = -8
MK_E_INVALID_ARG = -9
MK_E_UNSUPPORTED = -10
MK_E_RCCL = -11
MK_E_CORRUPT = -12
MK_COMM_ID_BYTES = 128

MK_ALGO_AUTO, MK_ALGO_AC, MK_ALGO_BNDMQ = 0, 1, 2
MK_FLAG_ASCII_CASE_INSENSITIVE = 1
MK_MODE_ANY, MK_MODE_HITS = 0, 1
MK_NUM_SUMMARY = 8
MK_SUM_HITS, MK_SUM_RECORDS_HIT, MK_SUM_RECORDS, MK_SUM_BASES, MK_SUM_CANDIDATES = 0, 1, 2, 3, 4

HIT_DTYPE = np.dtype([("rec", "<u8"), ("pat", "<u4"), ("pos", "<u4")])
ROW_DTYPE = np.dtype([("rec", "<u8"), ("pat", "<u4"), ("pos", "<u4"), ("file", "<u4"), ("_pad", "<u4")])

EXPORTS = [
    "mk_abi_version", "mk_last_error", "mk_device_count", "mk_read_kmers_from_text", "mk_parse_pattern_list",
    "mk_reverse_complement", "mk_canonical", "mk_recommend_aho_corasick", "mk_tune_q_value", "mk_generate_masks",
    "mk_free", "mk_matcher_create", "mk_matcher_create_ex", "mk_plan_geometry", "mk_matcher_destroy", "mk_matcher_algo", "mk_matcher_num_patterns",
    "mk_matcher_filter_info", "mk_matcher_class_info", "mk_matcher_filter_mode", "mk_scan_batch", "mk_scan_device", "mk_order_hits", "mk_order_hits_device", "mk_matcher_order_info", "mk_matcher_order_stats", "mk_matcher_kernel_name",
    "mk_matcher_launch_info", "mk_matcher_enable_timing", "mk_matcher_kernel_times", "mk_matcher_hint_hit_density", "mk_matcher_hint_record_lengths", "mk_matcher_set_fixed_record_length", "mk_matcher_check_device",
    "mk_extract_single", "mk_extract_fastq_text", "mk_upload_text_ahead", "mk_host_alloc", "mk_host_free", "mk_extract_paired", "mk_tag_records", "mk_tag_value", "mk_matcher_batch_times", "mk_synth_reads_device",
    "mk_synth_reads_host", "mk_synth_reads_device_range", "mk_reduce_counters", "mk_reduce_prepare", "mk_comm_available", "mk_comm_unique_id", "mk_comm_init",
    "mk_comm_reduce_counters", "mk_comm_size", "mk_comm_destroy",
    "mk_codec_create", "mk_codec_destroy", "mk_bgzf_deflate_bound", "mk_bgzf_deflate", "mk_bgzf_deflate_pieces", "mk_bgzf_inflate", "mk_bgzf_members", "mk_bgzf_eof",
    "mk_codec_times", "mk_codec_set_pass_limits", "mk_codec_set_inflate_kernel", "mk_codec_set_gzip_chunk", "mk_gzip_inflate_device", "mk_gzip_text_read", "mk_gzip_text_device", "mk_gzip_text_release", "mk_gzip_info", "mk_extract_fastq_bgzf", "mk_extract_window",
    "mk_tag_bam_window", "mk_matcher_set_bam_piece",
]


class MerkurioError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"[{code}] {msg}")
        self.code = code


class PatternError(MerkurioError):
    """src/pattern_matching.rs:28-36"""
    KINDS = {MK_E_EMPTY_PATTERN: "EmptyPattern", MK_E_INVALID_Q: "InvalidQGramLength",
             MK_E_PATTERN_TOO_LONG: "PatternTooLong"}

    @property
    def kind(self):
        return self.KINDS.get(self.code, "?")


class MatcherOptions(C.Structure):
    """mk_matcher_options (include/merkurio_hip.h): tuning / test hooks of mk_matcher_create_ex"""
    _fields_ = [("struct_size", C.c_uint32), ("force_stride", C.c_uint32), ("force_global_filter", C.c_uint32),
                ("gbloom_log2_blocks", C.c_uint32), ("tile_run", C.c_uint32), ("gbloom_kib", C.c_uint32),
                ("length_classes", C.c_uint32), ("force_split_len", C.c_uint32), ("force_stride2", C.c_uint32),
                ("force_q2", C.c_uint32)]

    def __init__(self, force_stride=0, force_global_filter=False, gbloom_log2_blocks=0, tile_run=0, gbloom_kib=0,
                 length_classes=0, force_split_len=0, force_stride2=0, force_q2=0, force_single_class=False):
        super().__init__(C.sizeof(MatcherOptions), int(force_stride), int(bool(force_global_filter)),
                         int(gbloom_log2_blocks), int(tile_run), int(gbloom_kib),
                         1 if force_single_class else int(length_classes), int(force_split_len), int(force_stride2), int(force_q2))


class WindowText(C.Structure):
    """mk_window_text (include/merkurio_hip.h): what mk_extract_fastq_bgzf hands back of a window's text"""
    _fields_ = [("text", C.c_void_p), ("text_cap", C.c_uint64), ("tail", C.c_void_p), ("tail_cap", C.c_uint64), ("kept", C.c_void_p),
                ("kept_cap", C.c_uint64), ("n_text", C.c_uint64), ("n_used", C.c_uint64), ("n_tail", C.c_uint64), ("n_kept_bytes", C.c_uint64)]


class WindowSource(C.Structure):
    """mk_window_source (include/merkurio_hip.h, v6): one input file's part of a window handed to mk_extract_window"""
    _fields_ = [("head", C.c_void_p), ("n_head", C.c_uint64), ("text", C.c_void_p), ("n_text", C.c_uint64), ("bgzf", C.c_void_p),
                ("n_bgzf", C.c_uint64), ("members", C.c_void_p), ("n_members", C.c_uint64), ("device_text", C.c_void_p), ("n_device_text", C.c_uint64), ("ends_at_record", C.c_uint32),
                ("reserved", C.c_uint32), ("rec_start", C.c_void_p), ("tail", C.c_void_p), ("tail_cap", C.c_uint64), ("kept", C.c_void_p),
                ("kept_cap", C.c_uint64), ("all", C.c_void_p), ("all_cap", C.c_uint64), ("n_window", C.c_uint64), ("n_used", C.c_uint64),
                ("n_tail", C.c_uint64), ("n_kept_bytes", C.c_uint64), ("n_rec_seen", C.c_uint64)]


MK_TEXT_FASTQ, MK_TEXT_FASTA = 0, 1


class BamWindow(C.Structure):
    """mk_bam_window (include/merkurio_hip.h, v7): a window of a BAM file handed to mk_tag_bam_window"""
    _fields_ = [("head", C.c_void_p), ("n_head", C.c_uint64), ("bgzf", C.c_void_p), ("n_bgzf", C.c_uint64), ("members", C.c_void_p), ("n_members", C.c_uint64),
                ("last", C.c_uint32), ("filter_matching", C.c_uint32), ("invert", C.c_uint32), ("tag", C.c_uint8 * 2), ("reserved", C.c_uint8 * 2),
                ("block_bytes", C.c_uint32), ("tail", C.c_void_p), ("tail_cap", C.c_uint64), ("out", C.c_void_p), ("out_cap", C.c_uint64),
                ("rows", C.c_void_p), ("rows_cap", C.c_uint64), ("row_name", C.c_void_p), ("names", C.c_void_p), ("names_cap", C.c_uint64),
                ("on_tail", C.c_void_p), ("on_tail_ctx", C.c_void_p),
                ("n_window", C.c_uint64), ("n_used", C.c_uint64), ("n_tail", C.c_uint64), ("n_rec", C.c_uint64), ("n_kept", C.c_uint64),
                ("out_text_bytes", C.c_uint64), ("out_len", C.c_uint64), ("n_rows", C.c_uint64), ("n_names_bytes", C.c_uint64), ("ms", C.c_float * 8)]


class Counters(C.Structure):
    _fields_ = [("nb_records_tot", C.c_uint64), ("nb_bases", C.c_uint64), ("nb_hits_tot", C.c_uint64 * 2),
                ("nb_records_hit", C.c_uint64 * 2), ("nb_records_extracted", C.c_uint64)]

    def as_dict(self, counts):
        return {"records": self.nb_records_tot, "bases": self.nb_bases,
                "hits": (self.nb_hits_tot[0], self.nb_hits_tot[1]),
                "records_hit": (self.nb_records_hit[0], self.nb_records_hit[1]),
                "extracted": self.nb_records_extracted, "pattern_hit_counts": counts.tolist()}


_lib = None


def lib_path():
    return _build.LIB_PATH


def _bind_to_torch_hip_runtime():
    """One HIP runtime per process.  PyTorch wheels bundle their own libamdhip64.so (same soname,
    libamdhip64.so.7, as /opt/rocm's); two runtimes in one process each try to own the GPU and the
    second one finds none.  If PyTorch is installed, load ITS runtime first so that
    libmerkurio_hip.so's DT_NEEDED binds to it and a later `import torch` shares it.  A host
    without PyTorch (the C++ CLI, a Rust host) uses the system runtime.
    Set MERKURIO_SYSTEM_HIP=1 to skip."""
    if os.environ.get("MERKURIO_SYSTEM_HIP") == "1":
        return None
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if not spec or not spec.submodule_search_locations:
        return None
    path = os.path.join(list(spec.submodule_search_locations)[0], "lib", "libamdhip64.so")
    if not os.path.exists(path):
        return None
    return C.CDLL(path, mode=C.RTLD_GLOBAL)


def load(build_if_missing=True):
    """Loads the shared library (building it in-tree with hipcc first if needed)."""
    global _lib
    if _lib is not None:
        return _lib
    _bind_to_torch_hip_runtime()
    path = os.environ.get("MERKURIO_LIB_PATH") or _build.LIB_PATH  # override: A/B runs against another build
    if path == _build.LIB_PATH and build_if_missing and (
            not os.path.exists(path) or os.environ.get("MERKURIO_REBUILD") == "1"):
        _build.build_lib()  # a present library is used as is (the driver's build() step keeps it fresh)
    if not os.path.exists(path):
        raise MerkurioError(MK_E_HIP, f"{path} is missing: run `python -m merkurio_amd.build`")
    L = C.CDLL(path)
    L.mk_last_error.restype = C.c_char_p
    L.mk_matcher_kernel_name.restype = C.c_char_p
    L.mk_matcher_kernel_name.argtypes = [C.c_void_p]
    L.mk_tune_q_value.restype = C.c_size_t
    L.mk_tune_q_value.argtypes = [C.c_size_t]
    L.mk_recommend_aho_corasick.argtypes = [C.c_size_t, C.c_size_t]
    L.mk_free.argtypes = [C.c_void_p]
    L.mk_matcher_destroy.argtypes = [C.c_void_p]
    L.mk_matcher_algo.argtypes = [C.c_void_p]
    L.mk_matcher_algo.restype = C.c_uint32
    L.mk_matcher_num_patterns.argtypes = [C.c_void_p]
    L.mk_matcher_num_patterns.restype = C.c_uint32
    L.mk_matcher_create.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_int32,
                                    C.POINTER(C.c_void_p)]
    L.mk_matcher_create_ex.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_int32,
                                       C.POINTER(MatcherOptions), C.POINTER(C.c_void_p)]
    L.mk_synth_reads_device_range.argtypes = [C.c_void_p, C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint32, C.c_uint32,
                                              C.c_void_p, C.c_void_p, C.c_void_p]
    L.mk_reduce_counters.argtypes = [C.POINTER(C.c_void_p), C.c_int, C.POINTER(C.c_void_p), C.c_size_t, C.c_void_p]
    L.mk_reduce_prepare.argtypes = [C.POINTER(C.c_void_p), C.c_int]
    L.mk_comm_unique_id.argtypes = [C.c_void_p]
    L.mk_comm_init.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int]
    L.mk_comm_reduce_counters.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]
    L.mk_comm_destroy.argtypes = [C.c_void_p]
    L.mk_comm_size.argtypes = [C.c_void_p, C.POINTER(C.c_int)]
    L.mk_scan_batch.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint32, C.c_void_p, C.c_void_p,
                                C.c_uint64, C.POINTER(C.c_uint64)]
    L.mk_scan_device.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64, C.c_uint32, C.c_void_p,
                                 C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p]
    L.mk_order_hits.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64]
    L.mk_order_hits_device.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p]
    L.mk_matcher_order_info.argtypes = [C.c_void_p, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
    L.mk_matcher_launch_info.argtypes = [C.c_void_p, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
    L.mk_matcher_enable_timing.argtypes = [C.c_void_p, C.c_uint32]
    L.mk_matcher_hint_hit_density.argtypes = [C.c_void_p, C.c_uint32]
    L.mk_matcher_hint_record_lengths.argtypes = [C.c_void_p, C.c_int]
    L.mk_matcher_set_fixed_record_length.argtypes = [C.c_void_p, C.c_uint32]
    L.mk_matcher_check_device.argtypes = [C.c_void_p, C.c_void_p]
    L.mk_matcher_batch_times.argtypes = [C.c_void_p, C.POINTER(C.c_float)]
    L.mk_matcher_kernel_times.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.POINTER(C.c_uint32)]
    L.mk_matcher_filter_info.argtypes = [C.c_void_p, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32),
                                         C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
    L.mk_matcher_filter_mode.argtypes = [C.c_void_p, C.POINTER(C.c_uint32), C.POINTER(C.c_uint64)]
    L.mk_plan_geometry.argtypes = [C.c_void_p, C.c_uint32, C.POINTER(MatcherOptions)] + [C.POINTER(C.c_uint32)] * 7
    L.mk_matcher_class_info.argtypes = [C.c_void_p, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), C.POINTER(C.c_uint32),
                                        C.POINTER(C.c_uint32)]
    L.mk_extract_single.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_int, C.c_int, C.c_void_p,
                                    C.c_void_p, C.c_uint64, C.POINTER(C.c_uint64), C.POINTER(Counters), C.c_void_p]
    L.mk_extract_fastq_text.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_int, C.c_int, C.c_uint64, C.POINTER(C.c_uint64), C.c_void_p,
                                        C.c_void_p, C.c_void_p, C.c_uint64, C.POINTER(C.c_uint64), C.POINTER(Counters), C.c_void_p,
                                        C.POINTER(C.c_uint32)]
    L.mk_upload_text_ahead.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64]
    L.mk_host_alloc.argtypes = [C.c_size_t, C.POINTER(C.c_void_p)]
    L.mk_host_free.argtypes = [C.c_void_p]
    L.mk_host_free.restype = None
    L.mk_codec_create.argtypes = [C.c_int, C.POINTER(C.c_void_p)]
    L.mk_codec_destroy.argtypes = [C.c_void_p]
    L.mk_codec_destroy.restype = None
    L.mk_bgzf_deflate_bound.argtypes = [C.c_uint64, C.c_uint32]
    L.mk_bgzf_deflate_bound.restype = C.c_uint64
    L.mk_bgzf_deflate.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint32, C.c_void_p, C.c_uint64, C.POINTER(C.c_uint64)]
    L.mk_bgzf_deflate_pieces.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint32, C.c_void_p, C.c_uint64, C.POINTER(C.c_uint64)]
    L.mk_bgzf_inflate.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64, C.POINTER(C.c_uint64)]
    L.mk_bgzf_members.argtypes = [C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64),
                                  C.POINTER(C.c_uint64)]
    L.mk_bgzf_eof.restype = C.POINTER(C.c_uint8 * 28)
    L.mk_codec_times.argtypes = [C.c_void_p, C.POINTER(C.c_float)]
    L.mk_codec_set_pass_limits.argtypes = [C.c_void_p, C.c_uint64, C.c_uint64]
    L.mk_codec_set_gzip_chunk.argtypes = [C.c_void_p, C.c_uint64]
    L.mk_codec_set_inflate_kernel.argtypes = [C.c_void_p, C.c_int]
    L.mk_gzip_inflate_device.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.POINTER(C.c_uint64), C.POINTER(C.c_uint32)]
    L.mk_gzip_text_read.argtypes = [C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64]
    L.mk_gzip_text_device.argtypes = [C.c_void_p, C.POINTER(C.c_uint64)]
    L.mk_gzip_text_device.restype = C.c_void_p
    L.mk_gzip_text_release.argtypes = [C.c_void_p]
    L.mk_gzip_info.argtypes = [C.c_void_p, C.POINTER(C.c_uint32), C.POINTER(C.c_float)]
    L.mk_extract_fastq_bgzf.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64, C.c_int,
                                        C.POINTER(WindowText), C.c_int, C.c_int, C.c_uint64, C.POINTER(C.c_uint64), C.c_void_p, C.c_void_p, C.c_void_p,
                                        C.c_uint64, C.POINTER(C.c_uint64), C.POINTER(Counters), C.c_void_p, C.POINTER(C.c_uint32)]
    L.mk_extract_window.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p, C.c_int, C.c_int, C.c_uint64, C.POINTER(C.c_uint64),
                                    C.c_void_p, C.c_void_p, C.c_uint64, C.POINTER(C.c_uint64), C.c_void_p, C.c_void_p, C.POINTER(C.c_uint32)]
    L.mk_matcher_order_stats.argtypes = [C.c_void_p, C.POINTER(C.c_uint64)]
    L.mk_tag_bam_window.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(BamWindow), C.c_int, C.POINTER(Counters), C.c_void_p, C.POINTER(C.c_uint32)]
    L.mk_matcher_set_bam_piece.argtypes = [C.c_void_p, C.c_uint32]
    L.mk_extract_paired.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p, C.c_uint64,
                                    C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_uint64, C.POINTER(C.c_uint64),
                                    C.POINTER(Counters), C.c_void_p]
    L.mk_tag_records.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_int, C.c_int, C.c_int, C.c_void_p,
                                 C.c_void_p, C.c_uint64, C.POINTER(C.c_uint64), C.POINTER(Counters), C.c_void_p,
                                 C.c_void_p, C.c_void_p, C.c_uint64]
    L.mk_tag_value.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_char_p, C.c_char_p, C.c_size_t,
                               C.POINTER(C.c_size_t)]
    L.mk_synth_reads_device.argtypes = [C.c_void_p, C.c_uint64, C.c_uint64, C.c_uint32, C.c_uint32, C.c_void_p,
                                        C.c_void_p, C.c_void_p]
    L.mk_synth_reads_host.argtypes = [C.c_void_p, C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint32, C.c_uint32,
                                      C.c_void_p, C.c_void_p]
    L.mk_read_kmers_from_text.argtypes = [C.c_void_p, C.c_size_t, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p),
                                          C.POINTER(C.c_uint32)]
    L.mk_parse_pattern_list.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_int, C.c_int, C.c_int, C.c_int,
                                        C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.POINTER(C.c_uint32)]
    L.mk_reverse_complement.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p]
    L.mk_canonical.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p]
    L.mk_generate_masks.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.POINTER(C.c_uint64)]
    _lib = L
    return L


def _check(rc):
    if rc == MK_OK:
        return
    msg = load().mk_last_error().decode(errors="replace")
    if rc in PatternError.KINDS:
        raise PatternError(rc, msg)
    raise MerkurioError(rc, msg)


def device_count():
    return load().mk_device_count()


# ------------------------------------------------------------------ packing helpers
def pack_list(items):
    items = [bytes(x) for x in items]
    off = np.zeros(len(items) + 1, dtype=np.uint32)
    if items:
        off[1:] = np.cumsum([len(x) for x in items])
    data = np.frombuffer(b"".join(items) + b"\0", dtype=np.uint8).copy()
    return data, off


def pack_records(seqs):
    seqs = [bytes(x) for x in seqs]
    off = np.zeros(len(seqs) + 1, dtype=np.uint64)
    if seqs:
        off[1:] = np.cumsum([len(x) for x in seqs], dtype=np.uint64)
    data = np.frombuffer(b"".join(seqs) + b"\0", dtype=np.uint8).copy()
    return data, off


def _take_list(pb, po, n):
    L = load()
    off = np.ctypeslib.as_array(C.cast(po, C.POINTER(C.c_uint32)), shape=(n.value + 1,)).copy()
    total = int(off[-1])
    data = C.string_at(pb, total) if total else b""
    L.mk_free(pb)
    L.mk_free(po)
    return [data[off[i]:off[i + 1]] for i in range(n.value)]


# ------------------------------------------------------------------ pattern preparation
def read_kmers_from_text(content: bytes):
    """helpers::read_kmers_from_file body (src/helpers.rs:152-156)"""
    pb, po, n = C.c_void_p(), C.c_void_p(), C.c_uint32()
    buf = np.frombuffer(bytes(content) + b"\0", dtype=np.uint8)
    _check(load().mk_read_kmers_from_text(buf.ctypes.data, len(content), C.byref(pb), C.byref(po), C.byref(n)))
    return _take_list(pb, po, n)


def read_kmers_from_file(path):
    with open(path, "rb") as f:
        return read_kmers_from_text(f.read())


def parse_pattern_list(kmer_file=None, kmer_seq=None, reverse_complement=False, canonical=False, lowercase=False,
                       uppercase=False):
    """helpers::parse_pattern_list (src/helpers.rs:76-133); the file has priority over kmer_seq"""
    if kmer_file is not None:
        raw = read_kmers_from_file(kmer_file)
    elif kmer_seq is not None:
        raw = [s.encode() if isinstance(s, str) else bytes(s) for s in kmer_seq]
    else:
        raise MerkurioError(MK_E_NO_PATTERNS, "No k-mer sequence provided.")
    data, off = pack_list(raw)
    pb, po, n = C.c_void_p(), C.c_void_p(), C.c_uint32()
    _check(load().mk_parse_pattern_list(data.ctypes.data, off.ctypes.data, len(raw), int(reverse_complement),
                                        int(canonical), int(lowercase), int(uppercase), C.byref(pb), C.byref(po),
                                        C.byref(n)))
    return _take_list(pb, po, n)


def reverse_complement(s: bytes) -> bytes:
    src = np.frombuffer(bytes(s) + b"\0", dtype=np.uint8)
    out = np.zeros(len(s) + 1, dtype=np.uint8)
    load().mk_reverse_complement(src.ctypes.data, len(s), out.ctypes.data)
    return out[:len(s)].tobytes()


def canonical(s: bytes) -> bytes:
    src = np.frombuffer(bytes(s) + b"\0", dtype=np.uint8)
    out = np.zeros(len(s) + 1, dtype=np.uint8)
    load().mk_canonical(src.ctypes.data, len(s), out.ctypes.data)
    return out[:len(s)].tobytes()


def recommend_aho_corasick(pattern_list) -> bool:
    return bool(load().mk_recommend_aho_corasick(len(pattern_list), max(len(p) for p in pattern_list)))


def tune_q_value(pattern) -> int:
    q = load().mk_tune_q_value(len(pattern))
    if q == 0:
        raise MerkurioError(MK_E_PATTERN_TOO_LONG, "Pattern length is too long for BNDMq.")
    return q


def generate_masks(pattern: bytes):
    src = np.frombuffer(bytes(pattern) + b"\0", dtype=np.uint8)
    masks = np.zeros(256, dtype=np.uint64)
    accept = C.c_uint64()
    _check(load().mk_generate_masks(src.ctypes.data, len(pattern), masks.ctypes.data, C.byref(accept)))
    return masks.tolist(), accept.value


def plan_geometry(lengths, options=None):
    """mk_plan_geometry: the filter geometry / length classes a matcher for patterns of these lengths would get"""
    lens = np.ascontiguousarray(lengths, dtype=np.uint32)
    opt = None if options is None else (options if isinstance(options, MatcherOptions) else MatcherOptions(**options))
    v = [C.c_uint32() for _ in range(7)]
    _check(load().mk_plan_geometry(lens.ctypes.data, len(lens), None if opt is None else C.byref(opt), *[C.byref(x) for x in v]))
    keys = ("q_gram", "stride", "in_lds", "split_len", "n_short", "q_gram2", "stride2")
    return dict(zip(keys, (x.value for x in v)))


# ------------------------------------------------------------------ matcher handle
class Matcher:
    """The matcher bundle the reference drivers hold (src/cmd_extract.rs:259): construction
    applies the reference's algorithm-selection rule; scans run on the GPU."""

    def __init__(self, patterns, algo=MK_ALGO_AUTO, q=0, case_insensitive=False, device=0, options=None):
        """options: None or a MatcherOptions / dict(force_stride=..., force_global_filter=...,
        gbloom_log2_blocks=...) -> mk_matcher_create_ex"""
        self.patterns = [p.encode() if isinstance(p, str) else bytes(p) for p in patterns]
        data, off = pack_list(self.patterns)
        self._h = C.c_void_p()
        flags = MK_FLAG_ASCII_CASE_INSENSITIVE if case_insensitive else 0
        if options is None:
            _check(load().mk_matcher_create(data.ctypes.data, off.ctypes.data, len(self.patterns), algo, q, flags,
                                            device, C.byref(self._h)))
        else:
            opt = options if isinstance(options, MatcherOptions) else MatcherOptions(**options)
            _check(load().mk_matcher_create_ex(data.ctypes.data, off.ctypes.data, len(self.patterns), algo, q, flags,
                                               device, C.byref(opt), C.byref(self._h)))
        self.device = device

    def close(self):
        if getattr(self, "_h", None):
            load().mk_matcher_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:  # interpreter shutdown: the module globals are already gone
            pass

    @property
    def handle(self):
        return self._h

    @property
    def algo(self):
        return load().mk_matcher_algo(self._h)

    @property
    def use_ac(self):
        return self.algo == MK_ALGO_AC

    @property
    def kernel_name(self):
        return load().mk_matcher_kernel_name(self._h).decode()

    def launch_info(self):
        g, b, l = C.c_uint32(), C.c_uint32(), C.c_uint32()
        _check(load().mk_matcher_launch_info(self._h, C.byref(g), C.byref(b), C.byref(l)))
        return {"grid_blocks": g.value, "block_threads": b.value, "lds_bytes": l.value}

    def batch_times_ms(self):
        """upload / device / download / host milliseconds of the last extract_single / tag_records call"""
        ms = (C.c_float * 4)()
        _check(load().mk_matcher_batch_times(self._h, ms))
        return {"upload": ms[0], "device": ms[1], "download": ms[2], "host": ms[3]}

    def order_info(self):
        """what the last mk_order_hits_device did: path 0 nothing, 1 record bins, 2 (record, end) bins, 3 library sort"""
        a, b, c = C.c_uint32(), C.c_uint32(), C.c_uint32()
        _check(load().mk_matcher_order_info(self._h, C.byref(a), C.byref(b), C.byref(c)))
        return {"path": a.value, "bins": b.value, "max_bin": c.value}

    def enable_timing(self, slots):
        _check(load().mk_matcher_enable_timing(self._h, slots))
        self._timing_slots = slots

    def kernel_times_ms(self):
        """durations of the retained scan-kernel launches (hipEvents on the launch stream)"""
        cap = max(1, getattr(self, "_timing_slots", 0))
        ms = np.zeros(cap, dtype=np.float32)
        n = C.c_uint32()
        _check(load().mk_matcher_kernel_times(self._h, ms.ctypes.data, cap, C.byref(n)))
        return ms[:n.value].tolist()

    def filter_info(self):
        q, s, e, tb = C.c_uint32(), C.c_uint32(), C.c_uint64(), C.c_uint64()
        _check(load().mk_matcher_filter_info(self._h, C.byref(q), C.byref(s), C.byref(e), C.byref(tb)))
        return {"q_gram": q.value, "stride": s.value, "entries": e.value, "table_bytes": tb.value}

    def class_info(self):
        """length classes (matcher.cpp: plan_classes): split_len 0 = one class"""
        a, b, c, d = C.c_uint32(), C.c_uint32(), C.c_uint32(), C.c_uint32()
        _check(load().mk_matcher_class_info(self._h, C.byref(a), C.byref(b), C.byref(c), C.byref(d)))
        return {"split_len": a.value, "n_short": b.value, "q_gram2": c.value, "stride2": d.value}

    def filter_mode(self):
        lds, fb = C.c_uint32(), C.c_uint64()
        _check(load().mk_matcher_filter_mode(self._h, C.byref(lds), C.byref(fb)))
        return {"in_lds": bool(lds.value), "filter_bytes": fb.value}

    def hint_hit_density(self, records_hit_per_1000):
        """performance hint for the next scan (never changes results): picks the load flavour of the kernel"""
        _check(load().mk_matcher_hint_hit_density(self._h, int(records_hit_per_1000)))

    # ---- batched scan, host buffers
    def scan(self, seqs, mode=MK_MODE_HITS, hits_cap=None):
        """-> (flags: np.bool_[n], hits: structured array in reference emission order)"""
        data, off = pack_records(seqs)
        return self.scan_packed(data, off, mode, hits_cap)

    def scan_packed(self, data, off, mode=MK_MODE_HITS, hits_cap=None):
        n = len(off) - 1
        flags = np.zeros(max(1, n), dtype=np.uint8)
        nh = C.c_uint64()
        cap = 1024 if hits_cap is None else hits_cap
        while True:
            hits = np.zeros(max(1, cap), dtype=HIT_DTYPE)
            rc = load().mk_scan_batch(self._h, data.ctypes.data, off.ctypes.data, n, mode, flags.ctypes.data,
                                      hits.ctypes.data, cap, C.byref(nh))
            if rc == MK_E_CAPACITY and hits_cap is None:
                cap = nh.value
                continue
            _check(rc)
            return flags[:n].astype(bool), hits[:nh.value]

    # ---- driver loops
    def _rows(self, call):
        cap = 4096
        while True:
            rows = np.zeros(cap, dtype=ROW_DTYPE)
            n_rows = C.c_uint64()
            rc = call(rows, cap, n_rows)
            if rc == MK_E_CAPACITY and n_rows.value > cap:
                cap = n_rows.value
                continue
            _check(rc)
            r = rows[:n_rows.value]
            return [(int(f), int(rec), int(p), int(pos)) for f, rec, p, pos in zip(r["file"], r["rec"], r["pat"], r["pos"])]

    def extract_single(self, seqs, logging=True, invert=False):
        data, off = pack_records(seqs)
        n = len(seqs)
        keep = np.zeros(max(1, n), dtype=np.uint8)
        res = {}

        def call(rows, cap, n_rows):
            res["c"], res["counts"] = Counters(), np.zeros(len(self.patterns), dtype=np.uint32)
            return load().mk_extract_single(self._h, data.ctypes.data, off.ctypes.data, n, int(logging), int(invert),
                                            keep.ctypes.data, rows.ctypes.data, cap, C.byref(n_rows),
                                            C.byref(res["c"]), res["counts"].ctypes.data)
        rows = self._rows(call)
        return keep[:n].astype(bool).tolist(), rows, res["c"].as_dict(res["counts"])

    def extract_fastq_text(self, text: bytes, logging=True, invert=False):
        """mk_extract_fastq_text: raw 4-line FASTQ text -> (status, rec_start list (n + 1), keep, rows, counters);
        status 1 = not plain FASTQ, the caller's own reader must take the window"""
        buf = np.frombuffer(text, dtype=np.uint8)
        cap = max(1, text.count(b"\n") // 4 + 2)
        n_rec, status = C.c_uint64(), C.c_uint32()
        rec_start = np.zeros(cap + 1, dtype=np.uint64)
        keep = np.zeros(cap, dtype=np.uint8)
        cnt = Counters()
        counts = np.zeros(len(self.patterns), dtype=np.uint32)
        rows = np.zeros(4096, dtype=ROW_DTYPE)
        n_rows = C.c_uint64()
        while True:
            c2, k2 = Counters(), np.zeros(len(self.patterns), dtype=np.uint32)
            rc = load().mk_extract_fastq_text(self._h, buf.ctypes.data if len(buf) else None, len(buf), int(logging), int(invert), cap,
                                              C.byref(n_rec), rec_start.ctypes.data, keep.ctypes.data, rows.ctypes.data, len(rows),
                                              C.byref(n_rows), C.byref(c2), k2.ctypes.data, C.byref(status))
            if rc == MK_E_CAPACITY and n_rows.value > len(rows):
                rows = np.zeros(n_rows.value, dtype=ROW_DTYPE)
                continue
            _check(rc)
            cnt, counts = c2, k2
            break
        n = n_rec.value
        out_rows = [(int(r["file"]), int(r["rec"]), int(r["pat"]), int(r["pos"])) for r in rows[:n_rows.value]] if logging else []
        return status.value, rec_start[:n + 1].tolist(), [bool(k) for k in keep[:n]], out_rows, cnt.as_dict(counts)

    def extract_fastq_bgzf(self, codec, head: bytes, blob: bytes, members, last, logging=True, invert=False, whole_text=True):
        """mk_extract_fastq_bgzf: head + the text of `members` (entries of bgzf_members(blob), out_off re-based to 0) ->
        (status, text, n_used, rec_start, keep, rows, counters); the members are inflated into the ingest buffer on the device.
        whole_text=False: `text` is a pair (tail, kept) -- the unfinished record at the window's end and the kept records' text"""
        mem = members.copy()
        if len(mem):
            mem["out_off"] -= mem["out_off"][0]
        n_text = len(head) + int(mem["isize"].sum()) if len(mem) else len(head)
        hb, bb = np.frombuffer(head, dtype=np.uint8), np.frombuffer(blob, dtype=np.uint8)
        text = np.zeros(max(1, n_text), dtype=np.uint8)
        tail = np.zeros(1 << 20, dtype=np.uint8)
        io = WindowText()
        if whole_text:
            io.text, io.text_cap = text.ctypes.data, n_text
        else:
            io.tail, io.tail_cap, io.kept, io.kept_cap = tail.ctypes.data, tail.size, text.ctypes.data, 64
        cap = max(1, n_text // 8 + 2)
        n_rec, status = C.c_uint64(), C.c_uint32()
        rec_start = np.zeros(cap + 1, dtype=np.uint64)
        keep = np.zeros(cap, dtype=np.uint8)
        rows = np.zeros(4096, dtype=ROW_DTYPE)
        n_rows = C.c_uint64()
        while True:
            c2, k2 = Counters(), np.zeros(len(self.patterns), dtype=np.uint32)
            rc = load().mk_extract_fastq_bgzf(self._h, codec._h, hb.ctypes.data if len(hb) else None, len(hb), bb.ctypes.data if len(bb) else None,
                                              len(bb), mem.ctypes.data if len(mem) else None, len(mem), int(bool(last)), C.byref(io), int(logging),
                                              int(invert), cap, C.byref(n_rec), rec_start.ctypes.data, keep.ctypes.data, rows.ctypes.data,
                                              len(rows), C.byref(n_rows), C.byref(c2), k2.ctypes.data, C.byref(status))
            if rc == MK_E_CAPACITY and n_rows.value > len(rows):
                rows = np.zeros(n_rows.value, dtype=ROW_DTYPE)
                continue
            if rc == MK_E_CAPACITY and not whole_text and io.n_kept_bytes > io.kept_cap:  # (the first call states the need: the retry path)
                io.kept_cap = io.n_kept_bytes
                continue
            _check(rc)
            break
        n = n_rec.value
        out_rows = [(int(r["file"]), int(r["rec"]), int(r["pat"]), int(r["pos"])) for r in rows[:n_rows.value]] if logging else []
        out_text = text[:io.n_text].tobytes() if whole_text else (tail[:io.n_tail].tobytes(), text[:io.n_kept_bytes].tobytes())
        return (status.value, out_text, io.n_used, rec_start[:n + 1].tolist(), [bool(k) for k in keep[:n]], out_rows, c2.as_dict(k2))

    def extract_window(self, sources, fmt=MK_TEXT_FASTQ, logging=True, invert=False, codec=None, want=("tail",)):
        """mk_extract_window.  sources: one dict per input file (one = single, two = paired) with keys
             head (bytes, default b""), text (bytes) OR blob + members (bgzf_members entries), ends_at_record (bool)
        want: which texts come back per source, any of "tail", "kept", "all".
        -> dict(status, n_rec, keep, rows, counters, sources=[dict(n_window, n_used, n_rec_seen, rec_start, tail, kept, all)])"""
        L = load()
        n_src = len(sources)
        arr = (WindowSource * n_src)()
        hold = []
        bound = 0
        for k, sd in enumerate(sources):
            S = arr[k]
            head = np.frombuffer(sd.get("head", b""), dtype=np.uint8)
            hold.append(head)
            S.head, S.n_head = (head.ctypes.data if head.size else None), head.size
            n_body = 0
            if "members" in sd:
                mem = sd["members"].copy()
                if len(mem):
                    mem["out_off"] -= mem["out_off"][0]
                blob = np.frombuffer(sd["blob"], dtype=np.uint8)
                hold += [mem, blob]
                S.bgzf, S.n_bgzf = (blob.ctypes.data if blob.size else None), blob.size
                S.members, S.n_members = (mem.ctypes.data if len(mem) else None), len(mem)
                n_body = int(mem["isize"].sum()) if len(mem) else 0
            elif "device_text" in sd:  # (pointer, bytes): a range of a text that lies on the device (mk_gzip_text_device)
                S.device_text, S.n_device_text = sd["device_text"]
                n_body = S.n_device_text
            else:
                text = np.frombuffer(sd.get("text", b""), dtype=np.uint8)
                hold.append(text)
                S.text, S.n_text = (text.ctypes.data if text.size else None), text.size
                n_body = text.size
            S.ends_at_record = int(bool(sd.get("ends_at_record", True)))
            bound = max(bound, head.size + n_body)
        cap = max(1, bound // (2 if fmt == MK_TEXT_FASTA else 8) + 2)
        bufs = []
        for k in range(n_src):
            S = arr[k]
            b = {"rec_start": np.zeros(cap + 1, dtype=np.uint64), "tail": np.zeros(64, dtype=np.uint8), "kept": np.zeros(64, dtype=np.uint8),
                 "all": np.zeros(max(1, bound), dtype=np.uint8)}
            bufs.append(b)
            S.rec_start = b["rec_start"].ctypes.data
            if "tail" in want:
                S.tail, S.tail_cap = b["tail"].ctypes.data, b["tail"].size
            if "kept" in want:
                S.kept, S.kept_cap = b["kept"].ctypes.data, b["kept"].size
            if "all" in want:
                S.all, S.all_cap = b["all"].ctypes.data, b["all"].size
        n_rec, status, n_rows = C.c_uint64(), C.c_uint32(), C.c_uint64()
        keep = np.zeros(cap, dtype=np.uint8)
        rows = np.zeros(4096, dtype=ROW_DTYPE)
        for _ in range(8):
            c2, k2 = Counters(), np.zeros(len(self.patterns), dtype=np.uint32)
            rc = L.mk_extract_window(self._h, codec._h if codec else None, fmt, n_src, arr, int(logging), int(invert), cap, C.byref(n_rec),
                                     keep.ctypes.data, rows.ctypes.data, len(rows), C.byref(n_rows), C.byref(c2), k2.ctypes.data, C.byref(status))
            if rc == MK_E_CAPACITY:  # the call states every need: grow what was too small and ask again
                grown = False
                if n_rows.value > len(rows):
                    rows = np.zeros(n_rows.value, dtype=ROW_DTYPE)
                    grown = True
                for k in range(n_src):
                    S, b = arr[k], bufs[k]
                    if "tail" in want and S.n_tail > S.tail_cap:
                        b["tail"] = np.zeros(S.n_tail, dtype=np.uint8)
                        S.tail, S.tail_cap = b["tail"].ctypes.data, b["tail"].size
                        grown = True
                    if "kept" in want and S.n_kept_bytes > S.kept_cap:
                        b["kept"] = np.zeros(S.n_kept_bytes, dtype=np.uint8)
                        S.kept, S.kept_cap = b["kept"].ctypes.data, b["kept"].size
                        grown = True
                if grown:
                    continue
            _check(rc)
            break
        n = n_rec.value
        out = {"status": status.value, "n_rec": n, "keep": [bool(x) for x in keep[:n]],
               "rows": [(int(r["file"]), int(r["rec"]), int(r["pat"]), int(r["pos"])) for r in rows[:n_rows.value]] if logging else [],
               "counters": c2.as_dict(k2), "sources": []}
        for k in range(n_src):
            S, b = arr[k], bufs[k]
            out["sources"].append({"n_window": S.n_window, "n_used": S.n_used, "n_rec_seen": S.n_rec_seen,
                                   "rec_start": b["rec_start"][:n + 1].tolist() if not status.value else [],
                                   "tail": b["tail"][:S.n_tail].tobytes() if "tail" in want else None,
                                   "kept": b["kept"][:S.n_kept_bytes].tobytes() if "kept" in want else None,
                                   "all": b["all"][:S.n_window].tobytes() if "all" in want else None})
        return out

    def extract_paired(self, seqs1, seqs2, logging=True, invert=False):
        d1, o1 = pack_records(seqs1)
        d2, o2 = pack_records(seqs2)
        n = len(seqs1)
        keep = np.zeros(max(1, n), dtype=np.uint8)
        res = {}

        def call(rows, cap, n_rows):
            res["c"], res["counts"] = Counters(), np.zeros(len(self.patterns), dtype=np.uint32)
            return load().mk_extract_paired(self._h, d1.ctypes.data, o1.ctypes.data, len(seqs1), d2.ctypes.data,
                                            o2.ctypes.data, len(seqs2), int(logging), int(invert), keep.ctypes.data,
                                            rows.ctypes.data, cap, C.byref(n_rows), C.byref(res["c"]),
                                            res["counts"].ctypes.data)
        rows = self._rows(call)
        return keep[:n].astype(bool).tolist(), rows, res["c"].as_dict(res["counts"])

    def tag_records(self, seqs, logging=True, filter_matching=False, invert=False):
        data, off = pack_records(seqs)
        n = len(seqs)
        keep = np.zeros(max(1, n), dtype=np.uint8)
        foff = np.zeros(n + 1, dtype=np.uint64)
        res = {"fcap": 4096}

        def call(rows, cap, n_rows):
            while True:
                res["c"], res["counts"] = Counters(), np.zeros(len(self.patterns), dtype=np.uint32)
                res["fpat"] = np.zeros(res["fcap"], dtype=np.uint32)
                rc = load().mk_tag_records(self._h, data.ctypes.data, off.ctypes.data, n, int(logging),
                                           int(filter_matching), int(invert), keep.ctypes.data, rows.ctypes.data, cap,
                                           C.byref(n_rows), C.byref(res["c"]), res["counts"].ctypes.data,
                                           foff.ctypes.data, res["fpat"].ctypes.data, res["fcap"])
                if rc == MK_E_CAPACITY and int(foff[n]) > res["fcap"]:
                    res["fcap"] = int(foff[n])
                    continue
                return rc
        rows = self._rows(call)
        found = [res["fpat"][int(foff[i]):int(foff[i + 1])].tolist() for i in range(n)]
        return keep[:n].astype(bool).tolist(), rows, res["c"].as_dict(res["counts"]), found

    def tag_bam_window(self, codec, head: bytes, blob: bytes, members, last, tag=b"km", logging=True, filter_matching=False, invert=False,
                       write=True, block_bytes=0, piece_bytes=0):
        """mk_tag_bam_window: head + the text of `members` (entries of bgzf_members(blob), out_off re-based to 0) as BAM records ->
        dict(status, n_window, n_used, n_rec, n_kept, tail, out (BGZF members of the tagged kept records), out_text_bytes, rows
        [(name, rec, pat, pos)], counters, ms)"""
        mem = members.copy()
        if len(mem):
            mem["out_off"] -= mem["out_off"][0]
        hb, bb = np.frombuffer(head, dtype=np.uint8), np.frombuffer(blob, dtype=np.uint8)
        n_text = len(head) + (int(mem["isize"].sum()) if len(mem) else 0)
        _check(load().mk_matcher_set_bam_piece(self._h, piece_bytes))
        w = BamWindow()
        w.head, w.n_head, w.bgzf, w.n_bgzf = (hb.ctypes.data if len(hb) else None), len(hb), (bb.ctypes.data if len(bb) else None), len(bb)
        w.members, w.n_members = (mem.ctypes.data if len(mem) else None), len(mem)
        w.last, w.filter_matching, w.invert, w.block_bytes = int(bool(last)), int(bool(filter_matching)), int(bool(invert)), block_bytes
        w.tag[0], w.tag[1] = tag[0], tag[1]
        tail = np.zeros(max(64, n_text), dtype=np.uint8)
        w.tail, w.tail_cap = tail.ctypes.data, tail.size
        out = np.zeros(1 << 16, dtype=np.uint8)
        rows = np.zeros(4096, dtype=ROW_DTYPE)
        row_name = np.zeros(4096, dtype=np.uint64)
        names = np.zeros(1 << 16, dtype=np.uint8)
        status = C.c_uint32()
        while True:
            if write:
                w.out, w.out_cap = out.ctypes.data, out.size
            w.rows, w.rows_cap, w.row_name, w.names, w.names_cap = rows.ctypes.data, len(rows), row_name.ctypes.data, names.ctypes.data, names.size
            c2, k2 = Counters(), np.zeros(len(self.patterns), dtype=np.uint32)
            rc = load().mk_tag_bam_window(self._h, codec._h, C.byref(w), int(logging), C.byref(c2), k2.ctypes.data, C.byref(status))
            if rc == MK_E_CAPACITY and (w.n_rows > len(rows) or w.n_names_bytes > names.size or w.out_len > out.size):
                if w.n_rows > len(rows):
                    rows, row_name = np.zeros(w.n_rows, dtype=ROW_DTYPE), np.zeros(w.n_rows, dtype=np.uint64)
                if w.n_names_bytes > names.size:
                    names = np.zeros(w.n_names_bytes, dtype=np.uint8)
                if w.out_len > out.size:
                    out = np.zeros(w.out_len, dtype=np.uint8)
                continue
            _check(rc)
            break
        nb = names[:w.n_names_bytes].tobytes()
        out_rows = []
        if logging and status.value == 0:
            for k in range(w.n_rows):
                a = int(row_name[k])
                out_rows.append((nb[a:nb.index(b"\0", a)], int(rows[k]["rec"]), int(rows[k]["pat"]), int(rows[k]["pos"])))
        return dict(status=status.value, n_window=w.n_window, n_used=w.n_used, n_rec=w.n_rec, n_kept=w.n_kept, tail=tail[:w.n_tail].tobytes(),
                    out=out[:w.out_len].tobytes(), out_text_bytes=w.out_text_bytes, rows=out_rows, counters=c2.as_dict(k2), ms=list(w.ms))

    def tag_value(self, found, existing=None) -> bytes:
        arr = np.asarray(list(found) + [0], dtype=np.uint32)
        cap = 256
        while True:
            out = C.create_string_buffer(cap)
            n = C.c_size_t()
            rc = load().mk_tag_value(self._h, arr.ctypes.data, len(found), existing, out, cap, C.byref(n))
            if rc == MK_E_CAPACITY:
                cap = n.value + 1
                continue
            _check(rc)
            return out.value


def reduce_counters(matchers, d_counter_ptrs, length):
    """mk_reduce_counters: in-place RCCL all-reduce (sum) of one device counter vector per handle
    (single process, one handle per GPU; handles sharing a device are pre-summed there).
    d_counter_ptrs: device addresses (ints).  Returns the sum as a numpy uint64 array."""
    n = len(matchers)
    hs = (C.c_void_p * n)(*[m.handle for m in matchers])
    ps = (C.c_void_p * n)(*[C.c_void_p(int(p)) for p in d_counter_ptrs])
    out = np.zeros(length, dtype=np.uint64)
    _check(load().mk_reduce_counters(hs, n, ps, length, out.ctypes.data))
    return out


# ------------------------------------------------------------------ reference-shaped single matchers
class BNDMq:
    """pattern_matching::BNDMq (src/pattern_matching.rs:42-153): one pattern, q-gram length q.
    q is validated like the reference and has no effect on results."""

    def __init__(self, pattern: bytes, q: int, device=0):
        if len(pattern) == 0:
            raise PatternError(MK_E_EMPTY_PATTERN, "Pattern is empty.")
        if q == 0:
            raise PatternError(MK_E_INVALID_Q, "Invalid q-gram length: 0. Must be between 1 and pattern length.")
        self._m = Matcher([pattern], MK_ALGO_BNDMQ, q, False, device)

    def find_all(self, text: bytes):
        _, hits = self._m.scan([text], MK_MODE_HITS)
        return hits["pos"].tolist()

    def find_iter(self, text: bytes):
        return iter(self.find_all(text))

    def find_match(self, text: bytes) -> bool:
        flags, _ = self._m.scan([text], MK_MODE_ANY)
        return bool(flags[0])


class AhoCorasick:
    """AhoCorasick::builder().kind(DFA).ascii_case_insensitive(ci).build(patterns) with the
    overlapping iterator (src/cmd_extract.rs:260-265,332)."""

    def __init__(self, patterns, ascii_case_insensitive=False, device=0):
        self._m = Matcher(patterns, MK_ALGO_AC, 0, ascii_case_insensitive, device)

    def find_overlapping_iter(self, text: bytes):
        """yields (pattern_index, start) in the crate's emission order"""
        _, hits = self._m.scan([text], MK_MODE_HITS)
        return iter(list(zip(hits["pat"].tolist(), hits["pos"].tolist())))

    def is_match(self, text: bytes) -> bool:
        flags, _ = self._m.scan([text], MK_MODE_ANY)
        return bool(flags[0])


# ---- BGZF codec (include/merkurio_hip.h v5; the reference's `bam` reader / writer, src/cmd_tag.rs:254-271,503-506) ----
MEMBER_DTYPE = np.dtype([("data_off", "<u8"), ("out_off", "<u8"), ("data_len", "<u4"), ("isize", "<u4"), ("crc", "<u4"), ("reserved", "<u4")])


def bgzf_members(data):
    """mk_bgzf_members: the member table of a BGZF byte string (host code) -> (members, consumed bytes, text bytes)"""
    L = load()
    buf = np.frombuffer(data, dtype=np.uint8)
    n, used, text = C.c_uint64(0), C.c_uint64(0), C.c_uint64(0)
    _check(L.mk_bgzf_members(buf.ctypes.data if buf.size else None, buf.size, None, 0, C.byref(n), C.byref(used), C.byref(text)))
    mem = np.zeros(n.value, dtype=MEMBER_DTYPE)
    _check(L.mk_bgzf_members(buf.ctypes.data if buf.size else None, buf.size, mem.ctypes.data, mem.size, C.byref(n), C.byref(used), C.byref(text)))
    return mem, used.value, text.value


def bgzf_eof():
    return bytes(load().mk_bgzf_eof().contents)


class Codec:
    """mk_codec: BGZF members deflated / inflated by the gfx950 kernels (no CPU path: without a device creation raises)"""

    def __init__(self, device=0):
        self._h = None
        self._L = load()
        h = C.c_void_p()
        _check(self._L.mk_codec_create(device, C.byref(h)))
        self._h = h

    def close(self):
        if self._h:
            self._L.mk_codec_destroy(self._h)
            self._h = None

    __del__ = close

    def deflate(self, data, block_bytes=0):
        src = np.frombuffer(data, dtype=np.uint8)
        out = np.empty(self._L.mk_bgzf_deflate_bound(src.size, block_bytes), dtype=np.uint8)
        n = C.c_uint64(0)
        t0 = time.perf_counter()
        rc = self._L.mk_bgzf_deflate(self._h, src.ctypes.data if src.size else None, src.size, block_bytes,
                                     out.ctypes.data if out.size else None, out.size, C.byref(n))
        self.last_call_s = time.perf_counter() - t0  # (the C call alone: not this wrapper's buffer and bytes-object copies)
        _check(rc)
        return out[:n.value].tobytes()

    def deflate_pieces(self, pieces, block_bytes=0):
        """mk_bgzf_deflate_pieces: the members of b"".join(pieces), which is never formed on the host"""
        arrs = [np.frombuffer(p, dtype=np.uint8) for p in pieces]
        ptrs = (C.c_void_p * len(arrs))(*[a.ctypes.data if a.size else None for a in arrs])
        sizes = (C.c_uint64 * len(arrs))(*[a.size for a in arrs])
        total = sum(a.size for a in arrs)
        out = np.empty(self._L.mk_bgzf_deflate_bound(total, block_bytes), dtype=np.uint8)
        n = C.c_uint64(0)
        _check(self._L.mk_bgzf_deflate_pieces(self._h, ptrs, sizes, len(arrs), block_bytes, out.ctypes.data if out.size else None, out.size,
                                              C.byref(n)))
        return out[:n.value].tobytes()

    def inflate(self, data, members=None, text_bytes=None):
        if members is None:
            members, used, text_bytes = bgzf_members(data)
            if used != len(data):
                raise MerkurioError(MK_E_CORRUPT, "trailing bytes behind the last whole BGZF member")
        src = np.frombuffer(data, dtype=np.uint8)
        out = np.empty(text_bytes, dtype=np.uint8)
        bad = C.c_uint64(0)
        t0 = time.perf_counter()
        rc = self._L.mk_bgzf_inflate(self._h, src.ctypes.data if src.size else None, src.size, members.ctypes.data if members.size else None,
                                     members.size, out.ctypes.data if out.size else None, out.size, C.byref(bad))
        self.last_call_s = time.perf_counter() - t0  # (the C call alone, first touch of the fresh output buffer included)
        _check(rc)
        return out.tobytes()

    def gunzip(self, gz):
        """mk_gzip_inflate_device + mk_gzip_text_read: one gzip member inflated in parallel pieces -> bytes, or None when the device
        did not take the file (the caller's zlib does then).  self.gzip_info = (pieces, ms of upload / search / pieces / resolution / CRC)"""
        src = np.frombuffer(gz, dtype=np.uint8)
        n, taken = C.c_uint64(0), C.c_uint32(0)
        t0 = time.perf_counter()
        _check(self._L.mk_gzip_inflate_device(self._h, src.ctypes.data if src.size else None, src.size, C.byref(n), C.byref(taken)))
        self.last_call_s = time.perf_counter() - t0
        seg, ms = C.c_uint32(0), (C.c_float * 5)()
        _check(self._L.mk_gzip_info(self._h, C.byref(seg), ms))
        self.gzip_info = (seg.value, tuple(round(x, 2) for x in ms))
        if not taken.value:
            return None
        out = np.empty(n.value, dtype=np.uint8)
        t0 = time.perf_counter()
        _check(self._L.mk_gzip_text_read(self._h, 0, out.ctypes.data if out.size else None, out.size))
        self.last_read_s = time.perf_counter() - t0
        return out.tobytes()

    def set_inflate_kernel(self, which=0):
        """0 = chosen per call, 1 = a lane per member, 2 = a wave per member (mk_codec_set_inflate_kernel)"""
        _check(self._L.mk_codec_set_inflate_kernel(self._h, which))

    def set_gzip_chunk(self, chunk_bytes=0):
        """nominal distance between two cuts of a gzip stream (mk_codec_set_gzip_chunk); 0 = by the stream's size"""
        _check(self._L.mk_codec_set_gzip_chunk(self._h, chunk_bytes))

    def set_pass_limits(self, deflate_members=0, inflate_text_bytes=0):
        _check(self._L.mk_codec_set_pass_limits(self._h, deflate_members, inflate_text_bytes))

    def times(self):
        ms = (C.c_float * 3)()
        _check(self._L.mk_codec_times(self._h, ms))
        return tuple(ms)
