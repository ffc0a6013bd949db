#!/usr/bin/env python3
"""first_call_probe, second half: the HIP calls mk_matcher_create makes, one by one, in a fresh process -- is the first
matcher's 0.18 s the runtime (context, first allocation) or the load of the library's code objects at its first launch?"""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("MERKURIO_SYSTEM_HIP", "1")
hip = C.CDLL("libamdhip64.so")
def t(label, f):
    a = time.perf_counter(); r = f(); print("%-46s %.3f s" % (label, time.perf_counter() - a)); return r
n = C.c_int()
t("hipGetDeviceCount (runtime start)", lambda: hip.hipGetDeviceCount(C.byref(n)))
t("hipSetDevice(0)", lambda: hip.hipSetDevice(0))
p = C.c_void_p()
t("first hipMalloc(1 MiB)", lambda: hip.hipMalloc(C.byref(p), 1 << 20))
s = C.c_void_p()
t("hipStreamCreateWithFlags", lambda: hip.hipStreamCreateWithFlags(C.byref(s), 1))
t("hipMemset 1 MiB (a runtime kernel) + sync", lambda: (hip.hipMemset(p, 0, 1 << 20), hip.hipDeviceSynchronize()))
from merkurio_amd import native as mk
L = t("dlopen libmerkurio_hip.so (17 code objects)", mk.load)
pats = [b"ACGTACGTACGTACGTACGTACGTACGTACG"]
m = t("first matcher (its first launch: build_tables)", lambda: mk.Matcher(pats))
m2 = t("second matcher", lambda: mk.Matcher(pats))
t("first scan (its TU's code object)", lambda: m.scan([b"ACGT" * 50] * 1000))
t("second scan", lambda: m.scan([b"ACGT" * 50] * 1000))
