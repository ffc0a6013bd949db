"""ctypes binding of the CPU oracle (oracle/mk_oracle.c).  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.
"""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
LIB_PATH = os.path.join(ORACLE_DIR, "_build", "libmk_oracle.so")

MKO_OK = 0
MKO_E_EMPTY_PATTERN = -1
MKO_E_INVALID_Q = -2
MKO_E_PATTERN_TOO_LONG = -3
MKO_E_NO_PATTERNS = -4
MKO_E_PAIR_MISMATCH = -6


def build_oracle(force=False):
    src = [os.path.join(ORACLE_DIR, f) for f in ("mk_oracle.c", "mk_oracle.h")]
    if (not force and os.path.exists(LIB_PATH)
            and all(os.path.getmtime(LIB_PATH) >= os.path.getmtime(s) for s in src)):
        return LIB_PATH
    subprocess.run(["make", "-C", ORACLE_DIR, "-s"], check=True)
    return LIB_PATH


class _Bndmq(C.Structure):
    _fields_ = [("m", C.c_size_t), ("q", C.c_size_t), ("masks", C.c_uint64 * 256), ("accept", C.c_uint64)]


class _Patterns(C.Structure):
    _fields_ = [("n", C.c_uint32), ("off", C.POINTER(C.c_uint32)), ("bytes", C.POINTER(C.c_uint8))]


class _Matcher(C.Structure):
    _fields_ = [("use_ac", C.c_int), ("ac", C.c_void_p), ("n_pat", C.c_uint32), ("bndmq", C.c_void_p)]


class _Row(C.Structure):
    _fields_ = [("file", C.c_uint8), ("rec", C.c_uint64), ("pat", C.c_uint32), ("pos", C.c_uint64)]


class _Rows(C.Structure):
    _fields_ = [("rows", C.POINTER(_Row)), ("n_rows", C.c_size_t), ("cap_rows", C.c_size_t)]


class _Counters(C.Structure):
    _fields_ = [("nb_records_tot", C.c_uint64), ("nb_bases", C.c_uint64), ("nb_hits_tot", C.c_uint64 * 2),
                ("nb_records_hit", C.c_uint64 * 2), ("nb_records_extracted", C.c_uint64)]


_lib = None


def lib():
    global _lib
    if _lib is None:
        build_oracle()
        L = C.CDLL(LIB_PATH)
        L.mko_bndmq_find_all.restype = C.c_size_t
        L.mko_bndm_find_all.restype = C.c_size_t
        L.mko_tune_q_value.restype = C.c_size_t
        L.mko_tune_q_value.argtypes = [C.c_size_t]
        L.mko_ac_find_overlapping.restype = C.c_size_t
        L.mko_ac_num_states.restype = C.c_size_t
        L.mko_ac_table_bytes.restype = C.c_size_t
        L.mko_complement.restype = C.c_uint8
        L.mko_tag_value.restype = C.c_void_p
        L.mko_recommend_aho_corasick.argtypes = [C.c_size_t, C.c_size_t]
        L.mko_select_aho_corasick.argtypes = [C.c_int, C.c_int, C.c_int, C.c_size_t, C.c_size_t]
        L.mko_free.argtypes = [C.c_void_p]
        L.mko_ac_free.argtypes = [C.c_void_p]
        _lib = L
    return _lib


def _u8(b):
    return (C.c_uint8 * max(1, len(b))).from_buffer_copy(bytes(b) if len(b) else b"\0")


def pack_list(items):
    """list[bytes] -> (concatenated bytes array, uint32 offsets array)"""
    items = [bytes(x) for x in items]
    off = np.zeros(len(items) + 1, dtype=np.uint32)
    if items:
        off[1:] = np.cumsum([len(x) for x in items])
    return b"".join(items), off


def pack_records(seqs):
    """list[bytes] -> (uint8 array, uint64 offsets array)"""
    seqs = [bytes(x) for x in seqs]
    off = np.zeros(len(seqs) + 1, dtype=np.uint64)
    if seqs:
        off[1:] = np.cumsum([len(x) for x in seqs], dtype=np.uint64)
    data = np.frombuffer(b"".join(seqs), dtype=np.uint8).copy()
    if data.size == 0:
        data = np.zeros(1, dtype=np.uint8)
    return data, off


# ---------------------------------------------------------------------------- matchers
def generate_masks(pattern: bytes):
    masks = (C.c_uint64 * 256)()
    accept = C.c_uint64(0)
    rc = lib().mko_generate_masks(_u8(pattern), C.c_size_t(len(pattern)), masks, C.byref(accept))
    return rc, list(masks), accept.value


class BNDMq:
    def __init__(self, pattern: bytes, q: int):
        self._b = _Bndmq()
        self.rc = lib().mko_bndmq_new(_u8(pattern), C.c_size_t(len(pattern)), C.c_size_t(q), C.byref(self._b))

    def find_match(self, text: bytes) -> bool:
        assert self.rc == 0
        return bool(lib().mko_bndmq_find_match(C.byref(self._b), _u8(text), C.c_size_t(len(text))))

    def find_all(self, text: bytes):
        assert self.rc == 0
        cap = max(16, len(text))
        out = (C.c_size_t * cap)()
        n = lib().mko_bndmq_find_all(C.byref(self._b), _u8(text), C.c_size_t(len(text)), out, C.c_size_t(cap))
        assert n <= cap
        return list(out[:n])


def bndm_find_all(pattern: bytes, text: bytes):
    cap = max(16, len(text))
    out = (C.c_size_t * cap)()
    n = lib().mko_bndm_find_all(_u8(pattern), C.c_size_t(len(pattern)), _u8(text), C.c_size_t(len(text)), out,
                                C.c_size_t(cap))
    return list(out[:n])


def tune_q_value(n):
    return lib().mko_tune_q_value(n)


def recommend_aho_corasick(patterns):
    return bool(lib().mko_recommend_aho_corasick(len(patterns), max(len(p) for p in patterns)))


def select_aho_corasick(case_insensitive, force_ac, q_given, patterns):
    return bool(lib().mko_select_aho_corasick(int(case_insensitive), int(force_ac), int(q_given), len(patterns),
                                              max(len(p) for p in patterns)))


def reverse_complement(s: bytes) -> bytes:
    out = (C.c_uint8 * max(1, len(s)))()
    lib().mko_reverse_complement(_u8(s), C.c_size_t(len(s)), out)
    return bytes(out[:len(s)])


def canonical(s: bytes) -> bytes:
    out = (C.c_uint8 * max(1, len(s)))()
    lib().mko_canonical(_u8(s), C.c_size_t(len(s)), out)
    return bytes(out[:len(s)])


def _patterns_to_list(p):
    res = []
    for i in range(p.n):
        a, b = p.off[i], p.off[i + 1]
        res.append(bytes(bytearray(p.bytes[a:b])))
    return res


def read_kmers_from_text(content: bytes):
    p = _Patterns()
    rc = lib().mko_read_kmers_from_text(_u8(content), C.c_size_t(len(content)), C.byref(p))
    if rc:
        return rc, None
    res = _patterns_to_list(p)
    lib().mko_patterns_free(C.byref(p))
    return 0, res


def parse_pattern_list(raw, reverse_complement=False, canonical=False, lowercase=False, uppercase=False):
    data, off = pack_list(raw)
    p = _Patterns()
    rc = lib().mko_parse_pattern_list(_u8(data), off.ctypes.data_as(C.POINTER(C.c_uint32)), C.c_uint32(len(raw)),
                                      int(reverse_complement), int(canonical), int(lowercase), int(uppercase),
                                      C.byref(p))
    if rc:
        return rc, None
    res = _patterns_to_list(p)
    lib().mko_patterns_free(C.byref(p))
    return 0, res


class AhoCorasick:
    def __init__(self, patterns, ascii_case_insensitive=False):
        data, off = pack_list(patterns)
        self._h = C.c_void_p()
        self.n_pat = len(patterns)
        rc = lib().mko_ac_build(_u8(data), off.ctypes.data_as(C.POINTER(C.c_uint32)), C.c_uint32(len(patterns)),
                                int(ascii_case_insensitive), C.byref(self._h))
        assert rc == 0

    def __del__(self):
        if getattr(self, "_h", None):
            lib().mko_ac_free(self._h)
            self._h = None

    def num_states(self):
        return lib().mko_ac_num_states(self._h)

    def table_bytes(self):
        return lib().mko_ac_table_bytes(self._h)

    def find_overlapping(self, text: bytes):
        """list of (pattern_idx, start) in the crate's emission order"""
        cap = 64
        while True:
            pat = np.zeros(cap, dtype=np.uint32)
            st = np.zeros(cap, dtype=np.uint64)
            n = lib().mko_ac_find_overlapping(self._h, _u8(text), C.c_size_t(len(text)),
                                              pat.ctypes.data_as(C.POINTER(C.c_uint32)),
                                              st.ctypes.data_as(C.POINTER(C.c_uint64)), C.c_size_t(cap))
            if n <= cap:
                return list(zip(pat[:n].tolist(), st[:n].tolist()))
            cap = n

    def is_match(self, text: bytes) -> bool:
        return bool(lib().mko_ac_is_match(self._h, _u8(text), C.c_size_t(len(text))))


# ---------------------------------------------------------------------------- drivers
class Matcher:
    """The (Option<AhoCorasick>, Vec<(String, BNDMq)>) bundle of cmd_extract.rs:259."""

    def __init__(self, patterns, use_ac, q=0, case_insensitive=False):
        self.patterns = [bytes(p) for p in patterns]
        data, off = pack_list(self.patterns)
        self._data = _u8(data)
        self._off = off
        self._pl = _Patterns(len(patterns), off.ctypes.data_as(C.POINTER(C.c_uint32)),
                             C.cast(self._data, C.POINTER(C.c_uint8)))
        self._m = _Matcher()
        self.use_ac = bool(use_ac)
        self.rc = lib().mko_matcher_build(C.byref(self._pl), int(use_ac), C.c_size_t(q), int(case_insensitive),
                                          C.byref(self._m))

    def __del__(self):
        if getattr(self, "_m", None) is not None:
            lib().mko_matcher_free(C.byref(self._m))
            self._m = None


def _rows_list(rows):
    res = [(rows.rows[i].file, rows.rows[i].rec, rows.rows[i].pat, rows.rows[i].pos) for i in range(rows.n_rows)]
    lib().mko_rows_free(C.byref(rows))
    return res


def _counters_dict(c, counts):
    return {
        "records": c.nb_records_tot, "bases": c.nb_bases, "hits": (c.nb_hits_tot[0], c.nb_hits_tot[1]),
        "records_hit": (c.nb_records_hit[0], c.nb_records_hit[1]), "extracted": c.nb_records_extracted,
        "pattern_hit_counts": counts.tolist(),
    }


def extract_single(m: Matcher, seqs, logging=True, invert=False):
    assert m.rc == 0
    data, off = pack_records(seqs)
    n = len(seqs)
    keep = np.zeros(max(1, n), dtype=np.uint8)
    rows, c = _Rows(), _Counters()
    counts = np.zeros(len(m.patterns), dtype=np.uint32)
    rc = lib().mko_extract_single(C.byref(m._m), data.ctypes.data_as(C.POINTER(C.c_uint8)),
                                  off.ctypes.data_as(C.POINTER(C.c_uint64)), C.c_uint64(n), int(logging),
                                  int(invert), keep.ctypes.data_as(C.POINTER(C.c_uint8)), C.byref(rows), C.byref(c),
                                  counts.ctypes.data_as(C.POINTER(C.c_uint32)))
    assert rc == 0
    return keep[:n].astype(bool).tolist(), _rows_list(rows), _counters_dict(c, counts)


def extract_single_packed(m: Matcher, data, off, logging=False, invert=False):
    """same as extract_single on an already packed batch (uint8 bytes, uint64 offsets);
    returns (keep: np.uint8[n], counters dict) -- used by bench.py's cpu_baseline leg"""
    assert m.rc == 0
    n = len(off) - 1
    keep = np.zeros(max(1, n), dtype=np.uint8)
    rows, c = _Rows(), _Counters()
    counts = np.zeros(len(m.patterns), dtype=np.uint32)
    rc = lib().mko_extract_single(C.byref(m._m), data.ctypes.data_as(C.POINTER(C.c_uint8)),
                                  off.ctypes.data_as(C.POINTER(C.c_uint64)), C.c_uint64(n), int(logging),
                                  int(invert), keep.ctypes.data_as(C.POINTER(C.c_uint8)),
                                  C.byref(rows) if logging else None, C.byref(c),
                                  counts.ctypes.data_as(C.POINTER(C.c_uint32)))
    assert rc == 0
    if logging:
        lib().mko_rows_free(C.byref(rows))
    return keep[:n], _counters_dict(c, counts)


def extract_paired(m: Matcher, seqs1, seqs2, logging=True, invert=False):
    assert m.rc == 0
    d1, o1 = pack_records(seqs1)
    d2, o2 = pack_records(seqs2)
    n = len(seqs1)
    keep = np.zeros(max(1, n), dtype=np.uint8)
    rows, c = _Rows(), _Counters()
    counts = np.zeros(len(m.patterns), dtype=np.uint32)
    rc = lib().mko_extract_paired(C.byref(m._m), d1.ctypes.data_as(C.POINTER(C.c_uint8)),
                                  o1.ctypes.data_as(C.POINTER(C.c_uint64)), C.c_uint64(len(seqs1)),
                                  d2.ctypes.data_as(C.POINTER(C.c_uint8)), o2.ctypes.data_as(C.POINTER(C.c_uint64)),
                                  C.c_uint64(len(seqs2)), int(logging), int(invert),
                                  keep.ctypes.data_as(C.POINTER(C.c_uint8)), C.byref(rows), C.byref(c),
                                  counts.ctypes.data_as(C.POINTER(C.c_uint32)))
    if rc:
        return rc, None, None
    return keep[:n].astype(bool).tolist(), _rows_list(rows), _counters_dict(c, counts)


def tag_records(m: Matcher, seqs, logging=True, filter_matching=False, invert=False):
    """returns keep, rows, counters, found (list of list of pattern idx, push order)"""
    assert m.rc == 0
    data, off = pack_records(seqs)
    n = len(seqs)
    keep = np.zeros(max(1, n), dtype=np.uint8)
    rows, c = _Rows(), _Counters()
    counts = np.zeros(len(m.patterns), dtype=np.uint32)
    foff = np.zeros(n + 1, dtype=np.uint64)
    fpat = C.POINTER(C.c_uint32)()
    rc = lib().mko_tag_records(C.byref(m._m), data.ctypes.data_as(C.POINTER(C.c_uint8)),
                               off.ctypes.data_as(C.POINTER(C.c_uint64)), C.c_uint64(n), int(logging),
                               int(filter_matching), int(invert), keep.ctypes.data_as(C.POINTER(C.c_uint8)),
                               C.byref(rows), C.byref(c), counts.ctypes.data_as(C.POINTER(C.c_uint32)),
                               foff.ctypes.data_as(C.POINTER(C.c_uint64)), C.byref(fpat))
    assert rc == 0
    found = [[fpat[k] for k in range(int(foff[i]), int(foff[i + 1]))] for i in range(n)]
    lib().mko_free(fpat)
    return keep[:n].astype(bool).tolist(), _rows_list(rows), _counters_dict(c, counts), found


def tag_value(patterns, found, existing=None) -> bytes:
    data, off = pack_list(patterns)
    buf = _u8(data)
    pl = _Patterns(len(patterns), off.ctypes.data_as(C.POINTER(C.c_uint32)), C.cast(buf, C.POINTER(C.c_uint8)))
    arr = np.asarray(found, dtype=np.uint32)
    if arr.size == 0:
        arr = np.zeros(1, dtype=np.uint32)
    p = lib().mko_tag_value(C.byref(pl), arr.ctypes.data_as(C.POINTER(C.c_uint32)), C.c_size_t(len(found)),
                            None if existing is None else C.c_char_p(existing))
    s = C.string_at(p)
    lib().mko_free(p)
    return s
