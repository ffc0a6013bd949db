/*
 * mk_oracle.c -- CPU restatement of MerKurio's pattern-matching hot path (TEST INFRASTRUCTURE).
 * See mk_oracle.h for the parity-pinning statement and the who-may-call rule.
 * Plain C11, no dependencies.  Every function cites the reference file:line it follows.
 */
#include "mk_oracle.h"

#include <assert.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------------------------------
 * src/pattern_preprocessing.rs:24-43  generate_masks
 * bit (m-1-j) of masks[p[j]]; accept = 1 << (m-1); m > 64 -> PatternTooLong.
 * (m == 0 never reaches this function in the reference: BNDMq::new rejects it first.)
 * ---------------------------------------------------------------------------------------- */
int mko_generate_masks(const uint8_t *pattern, size_t m, uint64_t masks[256], uint64_t *accept) {
    memset(masks, 0, 256 * sizeof(uint64_t));
    if (m > 64) return MKO_E_PATTERN_TOO_LONG;
    for (size_t j = 0; j < m; j++) masks[pattern[j]] |= (uint64_t)1 << (m - j - 1);
    *accept = m ? (uint64_t)1 << (m - 1) : 0;
    return MKO_OK;
}

/* src/pattern_matching.rs:61-78  BNDMq::new */
int mko_bndmq_new(const uint8_t *pattern, size_t m, size_t q, mko_bndmq *out) {
    if (m == 0) return MKO_E_EMPTY_PATTERN;
    if (q == 0 || q > m) return MKO_E_INVALID_Q;
    int rc = mko_generate_masks(pattern, m, out->masks, &out->accept);
    if (rc) return rc;
    out->m = m;
    out->q = q;
    return MKO_OK;
}

/* src/pattern_matching.rs:82-125 with on_match = |_| true (find_match, :128-130) */
int mko_bndmq_find_match(const mko_bndmq *b, const uint8_t *text, size_t n) {
    if (b->m > n) return 0;
    size_t step = b->m - b->q + 1;
    size_t i = step;
    while (i <= n - b->q + 1) {
        uint64_t state = b->masks[text[i - 1]];
        for (size_t ii = 0; ii + 1 < b->q; ii++) state &= b->masks[text[i + ii]] << (ii + 1);
        if (state != 0) {
            size_t j = i;
            size_t first = i - step;
            for (;;) {
                j -= 1;
                if (state >= b->accept) {
                    if (j > first)
                        i = j;
                    else
                        return 1;
                }
                assert(j >= 1);
                state = (state << 1) & b->masks[text[j - 1]];
                if (state == 0) break;
            }
        }
        i += step;
    }
    return 0;
}

/* src/pattern_matching.rs:133-209  find_iter + Matches::next, collected (find_all :151-153) */
size_t mko_bndmq_find_all(const mko_bndmq *b, const uint8_t *text, size_t n, size_t *out, size_t cap) {
    size_t count = 0;
    if (b->m > n) return 0;
    size_t step = b->m - b->q + 1;
    size_t i = step; /* iterator state, :137 */
    for (;;) {       /* one pass of this loop body == one call of Matches::next */
        int yielded = 0;
        while (i <= n - b->q + 1) {
            uint64_t state = b->masks[text[i - 1]];
            for (size_t ii = 0; ii + 1 < b->q; ii++) state &= b->masks[text[i + ii]] << (ii + 1);
            if (state != 0) {
                size_t j = i;
                size_t first = i - step;
                for (;;) {
                    j -= 1;
                    if (state >= b->accept) {
                        if (j > first) {
                            i = j;
                        } else {
                            i += step;
                            if (count < cap) out[count] = j;
                            count++;
                            yielded = 1;
                            break;
                        }
                    }
                    assert(j >= 1);
                    state = (state << 1) & b->masks[text[j - 1]];
                    if (state == 0) break;
                }
                if (yielded) break;
            }
            i += step;
        }
        if (!yielded) return count;
    }
}

/* src/pattern_matching.rs:213-225 */
size_t mko_tune_q_value(size_t len) {
    if (len <= 1) return 1;
    if (len <= 3) return 2;
    if (len <= 8) return 3;
    if (len <= 30) return 4;
    if (len <= 55) return 5;
    if (len <= 64) return 6;
    return 0;
}

/* src/pattern_matching.rs:265-298  legacy BNDM::find_all */
size_t mko_bndm_find_all(const uint8_t *pattern, size_t m, const uint8_t *text, size_t n, size_t *out,
                         size_t cap) {
    uint64_t masks[256], accept;
    size_t count = 0;
    if (m == 0 || mko_generate_masks(pattern, m, masks, &accept)) return 0;
    if (m > n) return 0;
    size_t i = 0;
    while (i <= n - m) {
        size_t j = m, last = m;
        uint64_t state = m == 64 ? ~(uint64_t)0 : (((uint64_t)1 << m) - 1);
        while (state != 0) {
            state &= masks[text[i + j - 1]];
            j -= 1;
            if (state & accept) {
                if (j > 0)
                    last = j;
                else {
                    if (count < cap) out[count] = i;
                    count++;
                    break;
                }
            }
            state <<= 1;
        }
        i += last;
    }
    return count;
}

/* src/helpers.rs:203-211 */
int mko_recommend_aho_corasick(size_t num_patterns, size_t max_len) {
    return (num_patterns >= 14 || max_len > 64) ? 1 : 0;
}

/* src/cmd_extract.rs:166-171, src/cmd_tag.rs:184-189 */
int mko_select_aho_corasick(int case_insensitive, int force_ac, int q_given, size_t num_patterns,
                            size_t max_len) {
    if (case_insensitive) return 1;
    if (!q_given && !force_ac) return mko_recommend_aho_corasick(num_patterns, max_len);
    return force_ac ? 1 : 0;
}

/* ------------------------------------------------------------------------------------------
 * [3P] needletail 0.6.3 sequence::complement / Sequence::reverse_complement / canonical,
 * as used at src/helpers.rs:103,117.
 * ---------------------------------------------------------------------------------------- */
uint8_t mko_complement(uint8_t c) {
    switch (c) {
    case 'a': return 't';
    case 'A': return 'T';
    case 'c': return 'g';
    case 'C': return 'G';
    case 'g': return 'c';
    case 'G': return 'C';
    case 't': return 'a';
    case 'T': return 'A';
    case 'r': return 'y';
    case 'y': return 'r';
    case 'k': return 'm';
    case 'm': return 'k';
    case 'b': return 'v';
    case 'v': return 'b';
    case 'd': return 'h';
    case 'h': return 'd';
    case 'R': return 'Y';
    case 'Y': return 'R';
    case 'K': return 'M';
    case 'M': return 'K';
    case 'B': return 'V';
    case 'V': return 'B';
    case 'D': return 'H';
    case 'H': return 'D';
    default: return c; /* s, w, n and every non-IUPAC byte pass through */
    }
}

void mko_reverse_complement(const uint8_t *in, size_t n, uint8_t *out) {
    for (size_t i = 0; i < n; i++) out[i] = mko_complement(in[n - 1 - i]);
}

void mko_canonical(const uint8_t *in, size_t n, uint8_t *out) {
    uint8_t *rc = (uint8_t *)malloc(n ? n : 1);
    mko_reverse_complement(in, n, rc);
    if (memcmp(rc, in, n) < 0)
        memcpy(out, rc, n);
    else
        memcpy(out, in, n);
    free(rc);
}

/* ------------------------------------------------------------------------------------------
 * Pattern list: src/helpers.rs:76-133
 * ---------------------------------------------------------------------------------------- */
typedef struct {
    const uint8_t *p;
    uint32_t len;
} slice_t;

static int slice_cmp(const void *a, const void *b) {
    const slice_t *x = (const slice_t *)a, *y = (const slice_t *)b;
    uint32_t n = x->len < y->len ? x->len : y->len;
    int c = n ? memcmp(x->p, y->p, n) : 0;
    if (c) return c;
    return (x->len > y->len) - (x->len < y->len);
}

static int patterns_from_slices(const slice_t *s, uint32_t n, mko_patterns *out) {
    size_t total = 0;
    for (uint32_t i = 0; i < n; i++) total += s[i].len;
    out->n = n;
    out->off = (uint32_t *)malloc((n + 1) * sizeof(uint32_t));
    out->bytes = (uint8_t *)malloc(total ? total : 1);
    if (!out->off || !out->bytes) return MKO_E_NOMEM;
    uint32_t o = 0;
    for (uint32_t i = 0; i < n; i++) {
        out->off[i] = o;
        if (s[i].len) memcpy(out->bytes + o, s[i].p, s[i].len);
        o += s[i].len;
    }
    out->off[n] = o;
    return MKO_OK;
}

int mko_parse_pattern_list(const uint8_t *in_bytes, const uint32_t *in_off, uint32_t n_in,
                           int reverse_complement, int canonical, int lowercase, int uppercase,
                           mko_patterns *out) {
    memset(out, 0, sizeof(*out));
    size_t in_total = in_off[n_in];
    /* working copy: originals (case converted) followed by reverse complements */
    uint32_t n_work = reverse_complement ? 2 * n_in : n_in;
    uint8_t *work = (uint8_t *)malloc((reverse_complement ? 2 : 1) * in_total + 1);
    slice_t *sl = (slice_t *)malloc((n_work ? n_work : 1) * sizeof(slice_t));
    if (!work || !sl) return MKO_E_NOMEM;
    memcpy(work, in_bytes, in_total);
    /* :92-96  lowercase wins over uppercase (ASCII folding; patterns are ASCII) */
    if (lowercase) {
        for (size_t i = 0; i < in_total; i++)
            if (work[i] >= 'A' && work[i] <= 'Z') work[i] |= 0x20;
    } else if (uppercase) {
        for (size_t i = 0; i < in_total; i++)
            if (work[i] >= 'a' && work[i] <= 'z') work[i] &= (uint8_t)~0x20;
    }
    for (uint32_t i = 0; i < n_in; i++) {
        sl[i].p = work + in_off[i];
        sl[i].len = in_off[i + 1] - in_off[i];
    }
    /* :99-109  append reverse complements */
    if (reverse_complement) {
        for (uint32_t i = 0; i < n_in; i++) {
            uint8_t *dst = work + in_total + in_off[i];
            mko_reverse_complement(sl[i].p, sl[i].len, dst);
            sl[n_in + i].p = dst;
            sl[n_in + i].len = sl[i].len;
        }
    }
    /* :112-121  canonical form (in place; same length) */
    if (canonical) {
        for (uint32_t i = 0; i < n_work; i++) {
            uint8_t *tmp = (uint8_t *)malloc(sl[i].len ? sl[i].len : 1);
            mko_canonical(sl[i].p, sl[i].len, tmp);
            memcpy((uint8_t *)sl[i].p, tmp, sl[i].len);
            free(tmp);
        }
    }
    /* :124-126  retain non-empty, sort_unstable, dedup */
    uint32_t k = 0;
    for (uint32_t i = 0; i < n_work; i++)
        if (sl[i].len) sl[k++] = sl[i];
    qsort(sl, k, sizeof(slice_t), slice_cmp);
    uint32_t u = 0;
    for (uint32_t i = 0; i < k; i++)
        if (u == 0 || slice_cmp(&sl[u - 1], &sl[i]) != 0) sl[u++] = sl[i];
    int rc = MKO_OK;
    if (u == 0)
        rc = MKO_E_NO_PATTERNS; /* :128-130 */
    else
        rc = patterns_from_slices(sl, u, out);
    free(sl);
    free(work);
    return rc;
}

static int is_ws(uint8_t c) { return c == ' ' || (c >= 9 && c <= 13); }

/* src/helpers.rs:139-163: str::lines() (splits on \n, strips one trailing \r), drop lines
 * that are empty or start with '#' / '>' BEFORE trimming, then trim(). */
int mko_read_kmers_from_text(const uint8_t *content, size_t len, mko_patterns *out) {
    memset(out, 0, sizeof(*out));
    size_t cap = 16, n = 0;
    slice_t *sl = (slice_t *)malloc(cap * sizeof(slice_t));
    size_t i = 0;
    while (i < len) {
        size_t e = i;
        while (e < len && content[e] != '\n') e++;
        size_t le = e;
        if (le > i && content[le - 1] == '\r') le--;
        if (le > i && content[i] != '#' && content[i] != '>') {
            size_t a = i, b = le;
            while (a < b && is_ws(content[a])) a++;
            while (b > a && is_ws(content[b - 1])) b--;
            if (n == cap) {
                cap *= 2;
                sl = (slice_t *)realloc(sl, cap * sizeof(slice_t));
            }
            sl[n].p = content + a;
            sl[n].len = (uint32_t)(b - a);
            n++;
        }
        i = e + 1;
    }
    int rc = n ? patterns_from_slices(sl, (uint32_t)n, out) : MKO_E_NO_PATTERNS; /* :158-160 */
    free(sl);
    return rc;
}

void mko_patterns_free(mko_patterns *p) {
    free(p->off);
    free(p->bytes);
    memset(p, 0, sizeof(*p));
}

/* ------------------------------------------------------------------------------------------
 * [3P] aho-corasick 1.1.3: AhoCorasick::builder().kind(DFA).ascii_case_insensitive(I)
 * .build(patterns) with the default MatchKind::Standard, and find_overlapping_iter.
 *
 * Published algorithm restated: trie (goto) over byte equivalence classes; failure links by
 * BFS; a state's match list = its own patterns in insertion (= pattern id) order followed
 * by the match list of its failure state; the DFA's transition for a missing goto edge is
 * the failure state's transition.  Overlapping search walks one transition per haystack
 * byte and, in every match state, yields each pattern of the state's list in order, with
 * end = current offset + 1 and start = end - len(pattern).
 * ascii_case_insensitive: ASCII letters compare equal regardless of case (both haystack and
 * patterns are folded before class lookup; equivalent to the crate adding both-case edges).
 * Anchor: tests/fixtures/extract/log.json (96 ordered hits, 14 mixed-length patterns).
 * ---------------------------------------------------------------------------------------- */
struct mko_ac {
    uint32_t n_pat;
    uint32_t *pat_len;
    uint32_t n_states;
    uint32_t n_classes;
    uint32_t stride_shift;
    uint16_t classes[256]; /* byte -> class; class 0 = "not in any pattern" */
    uint32_t *trans;      /* n_states << stride_shift, values are state ids */
    uint32_t *match_off;  /* n_states + 1 */
    uint32_t *match_pat;
};

static uint8_t fold(uint8_t c, int ci) { return (ci && c >= 'A' && c <= 'Z') ? (uint8_t)(c | 0x20) : c; }

int mko_ac_build(const uint8_t *pat_bytes, const uint32_t *pat_off, uint32_t n_pat,
                 int ascii_case_insensitive, mko_ac **out) {
    int ci = ascii_case_insensitive;
    mko_ac *ac = (mko_ac *)calloc(1, sizeof(mko_ac));
    if (!ac) return MKO_E_NOMEM;
    ac->n_pat = n_pat;
    ac->pat_len = (uint32_t *)malloc((n_pat ? n_pat : 1) * sizeof(uint32_t));
    /* byte classes: class 0 = bytes that occur in no pattern */
    int used[256] = {0};
    size_t total = pat_off[n_pat];
    for (size_t i = 0; i < total; i++) used[fold(pat_bytes[i], ci)] = 1;
    uint32_t nc = 1;
    uint16_t cls_of[256];
    memset(cls_of, 0, sizeof(cls_of));
    for (int b = 0; b < 256; b++)
        if (used[b]) cls_of[b] = (uint16_t)nc++;
    for (int b = 0; b < 256; b++) ac->classes[b] = cls_of[fold((uint8_t)b, ci)];
    ac->n_classes = nc;
    uint32_t sh = 0;
    while (((uint32_t)1 << sh) < nc) sh++;
    ac->stride_shift = sh;
    size_t stride = (size_t)1 << sh;

    /* trie */
    size_t max_states = total + 1;
    uint32_t *go = (uint32_t *)malloc(max_states * stride * sizeof(uint32_t));
    uint32_t *own_head = (uint32_t *)malloc(max_states * sizeof(uint32_t)); /* first own pattern */
    uint32_t *own_tail = (uint32_t *)malloc(max_states * sizeof(uint32_t));
    uint32_t *own_next = (uint32_t *)malloc((n_pat ? n_pat : 1) * sizeof(uint32_t));
    if (!go || !own_head || !own_tail || !own_next) return MKO_E_NOMEM;
    const uint32_t NONE = 0xFFFFFFFFu;
    uint32_t ns = 1;
    for (size_t c = 0; c < stride; c++) go[c] = NONE;
    own_head[0] = own_tail[0] = NONE;
    for (uint32_t p = 0; p < n_pat; p++) {
        uint32_t len = pat_off[p + 1] - pat_off[p];
        ac->pat_len[p] = len;
        uint32_t s = 0;
        for (uint32_t i = 0; i < len; i++) {
            uint16_t c = ac->classes[pat_bytes[pat_off[p] + i]];
            uint32_t t = go[(size_t)s * stride + c];
            if (t == NONE) {
                t = ns++;
                for (size_t k = 0; k < stride; k++) go[(size_t)t * stride + k] = NONE;
                own_head[t] = own_tail[t] = NONE;
                go[(size_t)s * stride + c] = t;
            }
            s = t;
        }
        own_next[p] = NONE;
        if (len == 0) continue; /* empty patterns never reach the builder (helpers.rs:124) */
        if (own_head[s] == NONE)
            own_head[s] = p;
        else
            own_next[own_tail[s]] = p;
        own_tail[s] = p;
    }
    ac->n_states = ns;
    ac->trans = (uint32_t *)malloc((size_t)ns * stride * sizeof(uint32_t));
    uint32_t *fail = (uint32_t *)malloc(ns * sizeof(uint32_t));
    uint32_t *queue = (uint32_t *)malloc(ns * sizeof(uint32_t));
    uint32_t *mcount = (uint32_t *)calloc(ns + 1, sizeof(uint32_t));
    if (!ac->trans || !fail || !queue || !mcount) return MKO_E_NOMEM;
    /* BFS: failure links + dense DFA */
    size_t qh = 0, qt = 0;
    fail[0] = 0;
    for (size_t c = 0; c < stride; c++) {
        uint32_t t = go[c];
        if (t != NONE && c < nc) {
            ac->trans[c] = t;
            fail[t] = 0;
            queue[qt++] = t;
        } else {
            ac->trans[c] = 0;
        }
    }
    while (qh < qt) {
        uint32_t s = queue[qh++];
        for (size_t c = 0; c < stride; c++) {
            uint32_t t = (c < nc) ? go[(size_t)s * stride + c] : NONE;
            if (t != NONE) {
                fail[t] = ac->trans[(size_t)fail[s] * stride + c];
                ac->trans[(size_t)s * stride + c] = t;
                queue[qt++] = t;
            } else {
                ac->trans[(size_t)s * stride + c] = ac->trans[(size_t)fail[s] * stride + c];
            }
        }
    }
    /* match lists: own (pattern-id order) ++ list(fail) ; sizes in BFS order */
    for (uint32_t p = own_head[0]; p != NONE; p = own_next[p]) mcount[0]++;
    for (size_t k = 0; k < qt; k++) {
        uint32_t s = queue[k];
        uint32_t n = 0;
        for (uint32_t p = own_head[s]; p != NONE; p = own_next[p]) n++;
        mcount[s] = n + mcount[fail[s]];
    }
    ac->match_off = (uint32_t *)malloc(((size_t)ns + 1) * sizeof(uint32_t));
    size_t tot = 0;
    for (uint32_t s = 0; s < ns; s++) {
        ac->match_off[s] = (uint32_t)tot;
        tot += mcount[s];
    }
    ac->match_off[ns] = (uint32_t)tot;
    ac->match_pat = (uint32_t *)malloc((tot ? tot : 1) * sizeof(uint32_t));
    {
        uint32_t w = ac->match_off[0];
        for (uint32_t p = own_head[0]; p != NONE; p = own_next[p]) ac->match_pat[w++] = p;
    }
    for (size_t k = 0; k < qt; k++) {
        uint32_t s = queue[k];
        uint32_t w = ac->match_off[s];
        for (uint32_t p = own_head[s]; p != NONE; p = own_next[p]) ac->match_pat[w++] = p;
        uint32_t f = fail[s];
        uint32_t fn = ac->match_off[f + 1] - ac->match_off[f];
        memcpy(ac->match_pat + w, ac->match_pat + ac->match_off[f], fn * sizeof(uint32_t));
    }
    free(go);
    free(own_head);
    free(own_tail);
    free(own_next);
    free(fail);
    free(queue);
    free(mcount);
    *out = ac;
    return MKO_OK;
}

void mko_ac_free(mko_ac *ac) {
    if (!ac) return;
    free(ac->pat_len);
    free(ac->trans);
    free(ac->match_off);
    free(ac->match_pat);
    free(ac);
}

size_t mko_ac_num_states(const mko_ac *ac) { return ac->n_states; }
size_t mko_ac_table_bytes(const mko_ac *ac) {
    return ((size_t)ac->n_states << ac->stride_shift) * sizeof(uint32_t);
}

size_t mko_ac_find_overlapping(const mko_ac *ac, const uint8_t *text, size_t n, uint32_t *out_pat,
                               uint64_t *out_start, size_t cap) {
    size_t count = 0;
    uint32_t s = 0;
    const uint32_t sh = ac->stride_shift;
    for (size_t i = 0; i < n; i++) {
        s = ac->trans[((size_t)s << sh) + ac->classes[text[i]]];
        uint32_t a = ac->match_off[s], b = ac->match_off[s + 1];
        for (uint32_t k = a; k < b; k++) {
            uint32_t p = ac->match_pat[k];
            if (count < cap) {
                out_pat[count] = p;
                out_start[count] = (uint64_t)(i + 1 - ac->pat_len[p]);
            }
            count++;
        }
    }
    return count;
}

int mko_ac_is_match(const mko_ac *ac, const uint8_t *text, size_t n) {
    uint32_t s = 0;
    const uint32_t sh = ac->stride_shift;
    for (size_t i = 0; i < n; i++) {
        s = ac->trans[((size_t)s << sh) + ac->classes[text[i]]];
        if (ac->match_off[s] != ac->match_off[s + 1]) return 1;
    }
    return 0;
}

/* ------------------------------------------------------------------------------------------
 * Matcher bundle: src/cmd_extract.rs:259-277 / src/cmd_tag.rs:234-252
 * ---------------------------------------------------------------------------------------- */
int mko_matcher_build(const mko_patterns *pl, int use_ac, size_t q, int case_insensitive,
                      mko_matcher *out) {
    memset(out, 0, sizeof(*out));
    out->use_ac = use_ac;
    out->n_pat = pl->n;
    if (use_ac) return mko_ac_build(pl->bytes, pl->off, pl->n, case_insensitive, &out->ac);
    out->bndmq = (mko_bndmq *)malloc((pl->n ? pl->n : 1) * sizeof(mko_bndmq));
    if (!out->bndmq) return MKO_E_NOMEM;
    for (uint32_t i = 0; i < pl->n; i++) {
        size_t len = pl->off[i + 1] - pl->off[i];
        size_t qq = q ? q : mko_tune_q_value(len);
        if (!q && qq == 0) return MKO_E_PATTERN_TOO_LONG; /* tune_q_value(..).unwrap() panics */
        int rc = mko_bndmq_new(pl->bytes + pl->off[i], len, qq, &out->bndmq[i]);
        if (rc) return rc;
    }
    return MKO_OK;
}

void mko_matcher_free(mko_matcher *m) {
    mko_ac_free(m->ac);
    free(m->bndmq);
    memset(m, 0, sizeof(*m));
}

/* ------------------------------------------------------------------------------------------
 * Driver loops
 * ---------------------------------------------------------------------------------------- */
static void rows_push(mko_rows *r, uint8_t file, uint64_t rec, uint32_t pat, uint64_t pos) {
    if (!r) return;
    if (r->n_rows == r->cap_rows) {
        r->cap_rows = r->cap_rows ? r->cap_rows * 2 : 64;
        r->rows = (mko_row *)realloc(r->rows, r->cap_rows * sizeof(mko_row));
    }
    mko_row *x = &r->rows[r->n_rows++];
    x->file = file;
    x->rec = rec;
    x->pat = pat;
    x->pos = pos;
}

void mko_rows_free(mko_rows *r) {
    free(r->rows);
    memset(r, 0, sizeof(*r));
}

typedef struct {
    uint32_t *pat;
    uint64_t *start;
    size_t *pos;
    size_t cap;
} scratch_t;

static void scratch_reserve(scratch_t *s, size_t n) {
    if (n <= s->cap) return;
    s->cap = n * 2 + 64;
    s->pat = (uint32_t *)realloc(s->pat, s->cap * sizeof(uint32_t));
    s->start = (uint64_t *)realloc(s->start, s->cap * sizeof(uint64_t));
    s->pos = (size_t *)realloc(s->pos, s->cap * sizeof(size_t));
}
static void scratch_free(scratch_t *s) {
    free(s->pat);
    free(s->start);
    free(s->pos);
}

/* all AC matches of one text into scratch; returns count */
static size_t ac_all(const mko_ac *ac, const uint8_t *t, size_t n, scratch_t *s) {
    size_t cnt = mko_ac_find_overlapping(ac, t, n, s->pat, s->start, s->cap);
    if (cnt > s->cap) {
        scratch_reserve(s, cnt);
        cnt = mko_ac_find_overlapping(ac, t, n, s->pat, s->start, s->cap);
    }
    return cnt;
}
static size_t bndmq_all(const mko_bndmq *b, const uint8_t *t, size_t n, scratch_t *s) {
    size_t cnt = mko_bndmq_find_all(b, t, n, s->pos, s->cap);
    if (cnt > s->cap) {
        scratch_reserve(s, cnt);
        cnt = mko_bndmq_find_all(b, t, n, s->pos, s->cap);
    }
    return cnt;
}

/* src/cmd_extract.rs:321-406 */
int mko_extract_single(const mko_matcher *m, const uint8_t *seq, const uint64_t *off, uint64_t n_rec,
                       int logging, int invert, uint8_t *keep, mko_rows *rows, mko_counters *c,
                       uint32_t *counts) {
    scratch_t sc = {0};
    scratch_reserve(&sc, 256);
    for (uint64_t r = 0; r < n_rec; r++) {
        const uint8_t *t = seq + off[r];
        size_t n = (size_t)(off[r + 1] - off[r]);
        int found = 0;
        if (logging) { /* :325-328 */
            c->nb_records_tot += 1;
            c->nb_bases += n;
        }
        if (m->use_ac) { /* :331-360 */
            if (!logging) {
                found = mko_ac_is_match(m->ac, t, n);
            } else {
                size_t cnt = ac_all(m->ac, t, n, &sc);
                for (size_t k = 0; k < cnt; k++) {
                    rows_push(rows, 0, r, sc.pat[k], sc.start[k]);
                    counts[sc.pat[k]] += 1;
                    c->nb_hits_tot[0] += 1;
                    found = 1;
                }
                if (found) c->nb_records_hit[0] += 1;
            }
        } else if (logging) { /* :365-387 */
            for (uint32_t idx = 0; idx < m->n_pat; idx++) {
                size_t cnt = bndmq_all(&m->bndmq[idx], t, n, &sc);
                for (size_t k = 0; k < cnt; k++) {
                    rows_push(rows, 0, r, idx, sc.pos[k]);
                    c->nb_hits_tot[0] += 1;
                }
                if (cnt) {
                    found = 1;
                    counts[idx] += 1;
                }
            }
            if (found) c->nb_records_hit[0] += 1;
        } else { /* :389-396 */
            for (uint32_t idx = 0; idx < m->n_pat; idx++)
                if (mko_bndmq_find_match(&m->bndmq[idx], t, n)) {
                    found = 1;
                    break;
                }
        }
        keep[r] = (uint8_t)((found != 0) != (invert != 0)); /* :400-405 */
        if (keep[r]) c->nb_records_extracted += 1;
    }
    scratch_free(&sc);
    return MKO_OK;
}

/* src/cmd_extract.rs:463-612 */
int mko_extract_paired(const mko_matcher *m, const uint8_t *seq1, const uint64_t *off1, uint64_t n_rec1,
                       const uint8_t *seq2, const uint64_t *off2, uint64_t n_rec2, int logging,
                       int invert, uint8_t *keep, mko_rows *rows, mko_counters *c, uint32_t *counts) {
    if (n_rec1 != n_rec2) return MKO_E_PAIR_MISMATCH; /* :465-468 / :608-612 */
    scratch_t sc = {0};
    scratch_reserve(&sc, 256);
    for (uint64_t r = 0; r < n_rec1; r++) {
        const uint8_t *t1 = seq1 + off1[r], *t2 = seq2 + off2[r];
        size_t n1 = (size_t)(off1[r + 1] - off1[r]), n2 = (size_t)(off2[r + 1] - off2[r]);
        int found = 0;
        if (logging) { /* :471-475 */
            c->nb_records_tot += 2;
            c->nb_bases += n1 + n2;
        }
        if (m->use_ac) { /* :478-537 */
            if (!logging) {
                found = mko_ac_is_match(m->ac, t1, n1) | mko_ac_is_match(m->ac, t2, n2);
            } else {
                uint64_t hit0 = 0, hit1 = 0;
                size_t cnt = ac_all(m->ac, t1, n1, &sc);
                for (size_t k = 0; k < cnt; k++) {
                    rows_push(rows, 0, r, sc.pat[k], sc.start[k]);
                    counts[sc.pat[k]] += 1;
                    hit0 = 1;
                    c->nb_hits_tot[0] += 1;
                    found = 1;
                }
                cnt = ac_all(m->ac, t2, n2, &sc);
                for (size_t k = 0; k < cnt; k++) {
                    rows_push(rows, 1, r, sc.pat[k], sc.start[k]);
                    counts[sc.pat[k]] += 1;
                    hit1 = 1;
                    c->nb_hits_tot[1] += 1;
                    found = 1;
                }
                c->nb_records_hit[0] += hit0;
                c->nb_records_hit[1] += hit1;
            }
        } else if (logging) { /* :542-587 */
            uint64_t hit0 = 0, hit1 = 0;
            for (uint32_t idx = 0; idx < m->n_pat; idx++) {
                size_t cnt = bndmq_all(&m->bndmq[idx], t1, n1, &sc);
                for (size_t k = 0; k < cnt; k++) {
                    rows_push(rows, 0, r, idx, sc.pos[k]);
                    c->nb_hits_tot[0] += 1;
                }
                int any1 = cnt != 0;
                cnt = bndmq_all(&m->bndmq[idx], t2, n2, &sc);
                for (size_t k = 0; k < cnt; k++) {
                    rows_push(rows, 1, r, idx, sc.pos[k]);
                    c->nb_hits_tot[1] += 1;
                }
                int any2 = cnt != 0;
                if (any1) {
                    found = 1;
                    hit0 = 1;
                    counts[idx] += 1;
                }
                if (any2) {
                    found = 1;
                    hit1 = 1;
                    counts[idx] += 1;
                }
            }
            c->nb_records_hit[0] += hit0;
            c->nb_records_hit[1] += hit1;
        } else { /* :589-596 */
            for (uint32_t idx = 0; idx < m->n_pat; idx++)
                if (mko_bndmq_find_match(&m->bndmq[idx], t1, n1) ||
                    mko_bndmq_find_match(&m->bndmq[idx], t2, n2)) {
                    found = 1;
                    break;
                }
        }
        keep[r] = (uint8_t)((found != 0) != (invert != 0)); /* :600-606 */
        if (keep[r]) c->nb_records_extracted += 2;
    }
    scratch_free(&sc);
    return MKO_OK;
}

/* src/cmd_tag.rs:387-467 */
int mko_tag_records(const mko_matcher *m, const uint8_t *seq, const uint64_t *off, uint64_t n_rec,
                    int logging, int filter_matching, int invert, uint8_t *keep, mko_rows *rows,
                    mko_counters *c, uint32_t *counts, uint64_t *found_off, uint32_t **found_pat) {
    scratch_t sc = {0};
    scratch_reserve(&sc, 256);
    size_t fcap = 256, fn = 0;
    uint32_t *fp = (uint32_t *)malloc(fcap * sizeof(uint32_t));
    for (uint64_t r = 0; r < n_rec; r++) {
        const uint8_t *t = seq + off[r];
        size_t n = (size_t)(off[r + 1] - off[r]);
        found_off[r] = fn;
        size_t nfound0 = fn;
        if (m->use_ac) { /* :392-414: always enumerates every hit */
            size_t cnt = ac_all(m->ac, t, n, &sc);
            for (size_t k = 0; k < cnt; k++) {
                if (fn == fcap) {
                    fcap *= 2;
                    fp = (uint32_t *)realloc(fp, fcap * sizeof(uint32_t));
                }
                fp[fn++] = sc.pat[k];
                if (logging) {
                    c->nb_hits_tot[0] += 1;
                    counts[sc.pat[k]] += 1;
                    rows_push(rows, 0, r, sc.pat[k], sc.start[k]);
                }
            }
        } else {
            for (uint32_t idx = 0; idx < m->n_pat; idx++) {
                int any;
                if (logging) { /* :417-434 */
                    size_t cnt = bndmq_all(&m->bndmq[idx], t, n, &sc);
                    for (size_t k = 0; k < cnt; k++) {
                        rows_push(rows, 0, r, idx, sc.pos[k]);
                        c->nb_hits_tot[0] += 1;
                    }
                    any = cnt != 0;
                    if (any) counts[idx] += 1;
                } else { /* :436-442: no early exit over patterns */
                    any = mko_bndmq_find_match(&m->bndmq[idx], t, n);
                }
                if (any) {
                    if (fn == fcap) {
                        fcap *= 2;
                        fp = (uint32_t *)realloc(fp, fcap * sizeof(uint32_t));
                    }
                    fp[fn++] = idx;
                }
            }
        }
        int has = fn != nfound0;
        if (logging) { /* :445-451 */
            c->nb_records_tot += 1;
            c->nb_bases += n;
            if (has) c->nb_records_hit[0] += 1;
        }
        /* :457-467 */
        keep[r] = (uint8_t)(filter_matching ? has : (invert ? !has : 1));
        if (keep[r]) c->nb_records_extracted += 1; /* bookkeeping only; tag prints no such line */
    }
    found_off[n_rec] = fn;
    *found_pat = fp;
    scratch_free(&sc);
    return MKO_OK;
}

static int str_cmp(const void *a, const void *b) { return strcmp(*(char *const *)a, *(char *const *)b); }

/* src/cmd_tag.rs:470-490 */
char *mko_tag_value(const mko_patterns *pl, const uint32_t *found_pat, size_t n_found, const char *existing) {
    size_t cap = n_found + 1, n = 0;
    if (existing)
        for (const char *p = existing; *p; p++) cap += (*p == ',');
    char **items = (char **)malloc((cap + 1) * sizeof(char *));
    for (size_t i = 0; i < n_found; i++) {
        uint32_t p = found_pat[i];
        uint32_t len = pl->off[p + 1] - pl->off[p];
        char *s = (char *)malloc(len + 1);
        memcpy(s, pl->bytes + pl->off[p], len);
        s[len] = 0;
        items[n++] = s;
    }
    if (existing && existing[0]) { /* empty existing value: nothing appended (:472) */
        const char *p = existing;
        for (;;) {
            const char *e = strchr(p, ',');
            size_t len = e ? (size_t)(e - p) : strlen(p);
            char *s = (char *)malloc(len + 1);
            memcpy(s, p, len);
            s[len] = 0;
            items[n++] = s;
            if (!e) break;
            p = e + 1;
        }
    }
    qsort(items, n, sizeof(char *), str_cmp);
    size_t total = 1;
    size_t u = 0;
    for (size_t i = 0; i < n; i++) {
        if (u && strcmp(items[u - 1], items[i]) == 0) {
            free(items[i]);
            continue;
        }
        items[u++] = items[i];
        total += strlen(items[u - 1]) + 1;
    }
    char *out = (char *)malloc(total + 1);
    size_t w = 0;
    for (size_t i = 0; i < u; i++) {
        if (i) out[w++] = ',';
        size_t l = strlen(items[i]);
        memcpy(out + w, items[i], l);
        w += l;
        free(items[i]);
    }
    out[w] = 0;
    free(items);
    return out;
}

void mko_free(void *p) { free(p); }
