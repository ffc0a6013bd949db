"""Independent naive enumerators (pure Python) used to cross-check the oracle itself.
Both reference matchers report exactly "all possibly-overlapping occurrences of every
pattern" (SURVEY.md §0.4); they differ only in emission order (§0.5)."""


def _fold(b, ci):
    return b.lower() if ci else b


def occurrences(pattern: bytes, text: bytes, ci=False):
    p, t = _fold(pattern, ci), _fold(text, ci)
    res, i = [], t.find(p)
    while i != -1:
        res.append(i)
        i = t.find(p, i + 1)
    return res


def ac_order(patterns, text: bytes, ci=False):
    """(pattern_idx, start) sorted as aho-corasick's overlapping DFA search emits them:
    end ascending; same end: longer pattern first (= start ascending); same span
    (only possible with ascii_case_insensitive duplicates): pattern id ascending."""
    hits = []
    for idx, p in enumerate(patterns):
        for s in occurrences(p, text, ci):
            hits.append((s + len(p), s, idx))
    hits.sort()
    return [(idx, s) for (_, s, idx) in hits]


def bndmq_order(patterns, text: bytes):
    """pattern-major, positions ascending (cmd_extract.rs:365-384)"""
    return [(idx, s) for idx, p in enumerate(patterns) for s in occurrences(p, text)]
