"""Randomised differential test: many small random pattern sets / record batches through the
GPU path vs the oracle, covering the generic (runtime-q) kernel variants, every stride the
geometry rule can pick, odd pattern lengths, mixed alphabets, -I, -r/-c pattern lists, forced
BNDMq / Aho-Corasick emission orders, the global-filter mode, length classes (a few much shorter
patterns next to the set: split by the rule or forced, every short-class stride / table kind, r04)
and the FASTQ-text entry point (mk_extract_fastq_text == mk_extract_single on the same reads, r04)."""
import os
import random

import pytest

import oracle_binding as ob

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def mk():
    from merkurio_amd import native
    native.load()
    if native.device_count() < 1:
        pytest.fail("no HIP device visible")
    return native


ALPHABETS = [b"ACGT", b"ACGTN", b"ACGTacgt", b"AC", b"ACDEFGHIKLMNPQRSTVWY", b"ACGTRYKMBVDHSWN"]


def _case(rnd):
    alpha = rnd.choice(ALPHABETS)
    n_pat = rnd.choice([1, 2, 5, 13, 14, 40, 300])
    base_len = rnd.choice([1, 2, 3, 4, 7, 8, 12, 15, 16, 17, 20, 21, 23, 24, 25, 28, 31, 32, 33, 40, 47, 48, 49, 64, 65, 90])
    mixed = rnd.random() < 0.3
    raw = []
    for _ in range(n_pat):
        L = max(1, base_len + (rnd.randrange(0, 9) if mixed else 0))
        raw.append(bytes(rnd.choice(alpha) for _ in range(L)))
    if base_len >= 15 and rnd.random() < 0.35:  # length classes: a few much shorter patterns next to the set
        for _ in range(rnd.choice([1, 1, 2, 5])):
            raw.append(bytes(rnd.choice(alpha) for _ in range(rnd.choice([1, 2, 3, 5, 8, 8, 9, 12, 14]))))
    recs = []
    for _ in range(rnd.choice([1, 20, 200])):
        n = rnd.choice([0, 1, base_len - 1 if base_len > 1 else 1, base_len, base_len + 1, 100, 151, 700])
        s = bytearray(rnd.choice(alpha) for _ in range(n))
        for _ in range(rnd.choice([0, 0, 1, 2])):
            p = rnd.choice(raw)
            if len(p) <= n:
                k = rnd.randrange(0, n - len(p) + 1)
                s[k:k + len(p)] = p
        recs.append(bytes(s))
    return raw, recs


# longer campaigns: MERKURIO_FUZZ_SEEDS=200 MERKURIO_FUZZ_BASE=5000 python -m pytest tests/test_gpu_fuzz.py -m gpu
@pytest.mark.parametrize("seed", range(int(os.environ.get("MERKURIO_FUZZ_SEEDS", "32"))))
def test_fuzz_vs_oracle(mk, seed):
    rnd = random.Random(int(os.environ.get("MERKURIO_FUZZ_BASE", "1000")) + seed)
    for it in range(60):
        raw, recs = _case(rnd)
        kw = dict(reverse_complement=rnd.random() < 0.3, canonical=False, lowercase=False, uppercase=rnd.random() < 0.1)
        if not kw["reverse_complement"] and rnd.random() < 0.2:
            kw["canonical"] = True
        ci = rnd.random() < 0.2
        rc, patterns = ob.parse_pattern_list(raw, **kw)
        assert rc == 0 and patterns == mk.parse_pattern_list(kmer_seq=raw, **kw)
        max_len = max(len(p) for p in patterns)
        choice = rnd.random()
        algo, q = mk.MK_ALGO_AUTO, 0
        if choice < 0.2:
            algo = mk.MK_ALGO_AC
        elif choice < 0.4 and max_len <= 64 and not ci:
            algo, q = mk.MK_ALGO_BNDMQ, rnd.choice([0, 1])
        options = dict(force_global_filter=True) if rnd.random() < 0.15 else None
        if rnd.random() < 0.3:  # length-class options; a set they cannot split is created by the rule instead
            options = dict(options or {}, length_classes=rnd.choice([1, 2, 2]), force_stride2=rnd.choice([0, 0, 1, 2, 4, 8]),
                           force_q2=rnd.choice([0, 0, 3, 6, 7, 8]))
        try:
            m = mk.Matcher(patterns, algo=algo, q=q, case_insensitive=ci, options=options)
        except mk.MerkurioError as e:
            if e.code != mk.MK_E_INVALID_ARG or not options or "length_classes" not in options:
                raise
            options = {k: v for k, v in options.items() if k == "force_global_filter"} or None
            m = mk.Matcher(patterns, algo=algo, q=q, case_insensitive=ci, options=options)
        use_ac = m.use_ac
        assert use_ac == (True if (ci or algo == mk.MK_ALGO_AC) else False if algo == mk.MK_ALGO_BNDMQ
                          else ob.select_aho_corasick(ci, False, False, patterns))
        om = ob.Matcher(patterns, use_ac, 0, ci)
        assert om.rc == 0
        logging, invert = rnd.random() < 0.8, rnd.random() < 0.3
        got = m.extract_single(recs, logging=logging, invert=invert)
        exp = ob.extract_single(om, recs, logging=logging, invert=invert)
        assert got == exp, (seed, it, patterns[:3], len(recs))
        if it % 3 == 0 and not any(b"\n" in r or b"\r" in r for r in recs):
            # the same reads as FASTQ text through the device's record index (ingest.hip)
            text = b"".join(b"@r%d x\n%s\n+\n%s\n" % (i, r, b"@" * len(r)) for i, r in enumerate(recs))
            status, rec_start, keep_t, rows_t, c_t = m.extract_fastq_text(text, logging=logging, invert=invert)
            assert status == 0 and (keep_t, rows_t, c_t) == exp and len(rec_start) == len(recs) + 1, (seed, it)
        if len(recs) >= 2:
            h = len(recs) // 2
            assert m.extract_paired(recs[:h], recs[h:2 * h], logging=True) == ob.extract_paired(om, recs[:h], recs[h:2 * h], logging=True)
        # tag: the keep/drop rule (-m / -v, src/cmd_tag.rs:457-467) with the same flags on both sides
        fm = rnd.random() < 0.5
        inv = (not fm) and rnd.random() < 0.5
        keep, rows, c, found = m.tag_records(recs, logging=logging, filter_matching=fm, invert=inv)
        keep_o, rows_o, c_o, found_o = ob.tag_records(om, recs, logging=logging, filter_matching=fm, invert=inv)
        assert keep == keep_o and rows == rows_o and c == c_o, (seed, it, fm, inv)
        assert found == [sorted(set(f)) for f in found_o]
