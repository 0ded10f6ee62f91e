/*
 * am_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * A plain-C, single-threaded CPU restatement of the reference's Aho-Corasick hot
 * path (channable/alfred-margaret v2.1.1.1).  Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may load this library; nothing under
 * alfred-margaret_amd/ links, imports or calls it.
 *
 * The reference is Haskell and no GHC exists in the build image or on the GPU box,
 * so oracle/_ref (a build of the reference itself) is impossible here.  Parity of
 * this restatement is pinned by the reference's own known-answer tests, committed as
 * data under tests/golden/ (tests/Data/Text/AhoCorasickSpec.hs, Utf8Spec.hs and
 * README.md of the reference; see tests/golden/README.md for the line numbers).
 * Simple lowercase mapping: Unicode 14.0 table written by tools/gen_unicode_lower_node.js from
 * node/ICU data -- a different program and a different source than the product's table
 * (tools/gen_unicode_lower.py, Python's unicodedata; tests/test_unicode_lower.py compares them).
 * The reference defers to GHC base's Data.Char.toLower, whose Unicode version is not fixed by
 * the reference -> code points added after 14.0 (Unicode 16's Garay etc.): parity unpinned.
 *
 * Every function cites the reference lines it follows, relative to /root/reference.
 * Values `v` of the reference's `AcMachine v` are restated as uint32 handles that
 * the caller maps to payloads (needle index in every caller here).
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------ utf8 + lowercase */

typedef struct { uint32_t from, to; } lower_pair;
static const lower_pair k_lower[] = {
#include "unicode_lower_tbl.inc"
};
#define N_LOWER (sizeof(k_lower) / sizeof(k_lower[0]))

/* src/Data/Text/Utf8.hs:131-135 toLowerAscii, :145-151 lowerCodePoint
 * (non-ASCII: Data.Char.toLower of GHC base = simple lowercase mapping). */
/* The table Data.Char.toLower uses follows the GHC that builds the reference (alfred-margaret.cabal:69, .semaphore/semaphore.yml:92),
 * so tests may give the oracle the caller's pairs -- the same data libam's am_automaton_create_ex takes: orc_set_lower_table(from, to, n)
 * replaces the built-in Unicode 14.0 table process-wide until orc_set_lower_table(NULL, NULL, 0).  `from` must be sorted ascending. */
static const uint32_t* g_lower_from = NULL;
static const uint32_t* g_lower_to = NULL;
static size_t g_lower_n = 0;
void orc_set_lower_table(const uint32_t* from, const uint32_t* to, size_t n)
{
    free((void*)g_lower_from); free((void*)g_lower_to);
    g_lower_from = g_lower_to = NULL; g_lower_n = 0;
    if (!from || !to || !n) return;
    uint32_t* f = (uint32_t*)malloc(n * sizeof(uint32_t)); uint32_t* t = (uint32_t*)malloc(n * sizeof(uint32_t));
    memcpy(f, from, n * sizeof(uint32_t)); memcpy(t, to, n * sizeof(uint32_t));
    g_lower_from = f; g_lower_to = t; g_lower_n = n;
}

uint32_t orc_lower_code_point(uint32_t cp)
{
    if (cp < 128) return (cp >= 'A' && cp <= 'Z') ? cp + 0x20 : cp;
    if (g_lower_n) {
        size_t lo = 0, hi = g_lower_n;
        while (lo < hi) {
            size_t mid = (lo + hi) / 2;
            if (g_lower_from[mid] < cp) lo = mid + 1; else hi = mid;
        }
        return (lo < g_lower_n && g_lower_from[lo] == cp) ? g_lower_to[lo] : cp;
    }
    size_t lo = 0, hi = N_LOWER;
    while (lo < hi) {
        size_t mid = (lo + hi) / 2;
        if (k_lower[mid].from < cp) lo = mid + 1; else hi = mid;
    }
    return (lo < N_LOWER && k_lower[lo].from == cp) ? k_lower[lo].to : cp;
}

/* src/Data/Text/Utf8.hs:337-350 unsafeIndexCodePoint' / decodeN, :183-218 decode1..4.
 * The Haskell reads of cu1..cu3 are lazy; here they are guarded by `end` so the
 * oracle never reads past the buffer on (invalid) truncated input. */
static inline uint32_t decode_at(const uint8_t* d, size_t idx, size_t end, size_t* units)
{
    uint32_t cu0 = d[idx];
    uint32_t cu1 = idx + 1 < end ? d[idx + 1] : 0;
    uint32_t cu2 = idx + 2 < end ? d[idx + 2] : 0;
    uint32_t cu3 = idx + 3 < end ? d[idx + 3] : 0;
    if (cu0 < 0xc0) { *units = 1; return cu0; }
    if (cu0 < 0xe0) { *units = 2; return ((cu0 & 0x1f) << 6) | (cu1 & 0x3f); }
    if (cu0 < 0xf0) { *units = 3; return ((cu0 & 0xf) << 12) | ((cu1 & 0x3f) << 6) | (cu2 & 0x3f); }
    *units = 4;
    return ((cu0 & 0x7) << 18) | ((cu1 & 0x3f) << 12) | ((cu2 & 0x3f) << 6) | (cu3 & 0x3f);
}

/* src/Data/Text/Utf8.hs:154-160 unicode2utf8 */
static size_t encode_utf8(uint32_t c, uint8_t* out)
{
    if (c < 0x80) { out[0] = (uint8_t)c; return 1; }
    if (c < 0x800) { out[0] = 0xc0 | (c >> 6); out[1] = 0x80 | (c & 0x3f); return 2; }
    if (c < 0x10000) { out[0] = 0xe0 | (c >> 12); out[1] = 0x80 | ((c >> 6) & 0x3f); out[2] = 0x80 | (c & 0x3f); return 3; }
    out[0] = 0xf0 | (c >> 18); out[1] = 0x80 | ((c >> 12) & 0x3f); out[2] = 0x80 | ((c >> 6) & 0x3f); out[3] = 0x80 | (c & 0x3f);
    return 4;
}

/* src/Data/Text/Utf8.hs:138-140 lowerUtf8 = Text.map lowerCodePoint.
 * Returns malloc'd bytes; *out_len = new length (can differ from len). */
uint8_t* orc_lower_utf8(const uint8_t* d, size_t len, size_t* out_len)
{
    uint8_t* out = (uint8_t*)malloc(len * 4 + 4);   /* worst case never exceeds 4 bytes per input byte */
    size_t o = 0, i = 0;
    while (i < len) {
        size_t units;
        uint32_t cp = decode_at(d, i, len, &units);
        o += encode_utf8(orc_lower_code_point(cp), out + o);
        i += units;
    }
    *out_len = o;
    return out;
}

/* Text.length: number of code points (used at Replacer.hs:113). */
static size_t count_code_points(const uint8_t* d, size_t len)
{
    size_t n = 0, i = 0;
    while (i < len) { size_t u; (void)decode_at(d, i, len, &u); i += u; n++; }
    return n;
}

/* src/Data/Text/Utf8.hs:256-276 skipCodePointsBackwards.  Returns -1 where the
 * reference calls `error`. `d` points at the first byte of the slice (off applied). */
int64_t orc_skip_code_points_backwards(const uint8_t* d, size_t len, int64_t index0, int64_t n0)
{
    if (index0 >= (int64_t)len) return -1;
    int64_t index = index0, n = n0;
    for (;;) {
        if (index >= 0 && (d[index] & 0xC0) == 0x80) { index--; continue; }  /* atTrailingByte */
        if (n == 0) { if (index < 0) return -1; return index; }
        if (index < 0) return -1;   /* the reference would read before the array here */
        index--; n--;
    }
}

/* ------------------------------------------------------------------ automaton construction */

typedef struct { int32_t cp; int32_t next; } edge_t;
typedef struct { edge_t* e; uint32_t n, cap; } edges_t;      /* IntMap State, ascending key order */
typedef struct { uint32_t* v; uint32_t n, cap; } vals_t;

typedef struct orc_machine {
    /* packed form: Automaton.hs:108-123 AcMachine */
    uint64_t* transitions; size_t n_transitions;    /* machineTransitions */
    uint32_t* offsets;                              /* machineOffsets, n_states+1 entries (:170 scanl) */
    uint64_t root_ascii[128];                       /* machineRootAsciiTransitions */
    size_t n_states;
    uint32_t* values; uint64_t* values_off;         /* machineValues: values[values_off[s] .. values_off[s+1]) */
    uint32_t* fallback;                             /* kept for tests */
} orc_machine;

#define WILDCARD 0x200000ull                         /* Automaton.hs:130-131 */
static inline uint64_t new_transition(uint32_t cp, uint32_t st) { return ((uint64_t)st << 32) | cp; }      /* :147-153 */
static inline uint64_t new_wildcard(uint32_t st) { return ((uint64_t)st << 32) | WILDCARD; }              /* :155-160 */
static inline int t_is_wildcard(uint64_t t) { return (t & WILDCARD) == WILDCARD; }                         /* :144-145 */
static inline uint32_t t_state(uint64_t t) { return (uint32_t)(t >> 32); }                                 /* :139-140 */
static inline uint32_t t_code(uint64_t t) { return (uint32_t)(t & 0x1fffff); }                             /* :135-136 */

static int edges_find(const edges_t* es, int32_t cp)
{
    uint32_t lo = 0, hi = es->n;
    while (lo < hi) { uint32_t mid = (lo + hi) / 2; if (es->e[mid].cp < cp) lo = mid + 1; else hi = mid; }
    return (lo < es->n && es->e[lo].cp == cp) ? (int)lo : -1;
}
static void edges_insert(edges_t* es, int32_t cp, int32_t next)
{
    if (es->n == es->cap) { es->cap = es->cap ? es->cap * 2 : 2; es->e = (edge_t*)realloc(es->e, es->cap * sizeof(edge_t)); }
    uint32_t i = es->n++;
    while (i > 0 && es->e[i - 1].cp > cp) { es->e[i] = es->e[i - 1]; i--; }
    es->e[i].cp = cp; es->e[i].next = next;
}

/* Automaton.hs:309-332 foldBreadthFirst: returns the visiting order.  The amortised
 * queue prepends `extra` (ascending) to the reversed backlog, so siblings are visited in
 * DESCENDING key order; the level order is what the algorithms below rely on. */
static uint32_t* bfs_order(const edges_t* trans, size_t n_states)
{
    uint32_t* order = (uint32_t*)malloc(n_states * sizeof(uint32_t));
    uint32_t* front = (uint32_t*)malloc(n_states * sizeof(uint32_t));
    uint32_t* rev = (uint32_t*)malloc(n_states * sizeof(uint32_t));   /* reverse(revBacklog) */
    size_t n_order = 0, nf = 1, fi = 0, nr = 0;
    front[0] = 0;
    for (;;) {
        if (fi == nf) {                     /* go [] revBacklog = go (reverse revBacklog) [] */
            if (nr == 0) break;
            memcpy(front, rev, nr * sizeof(uint32_t)); nf = nr; fi = 0; nr = 0;
            continue;
        }
        uint32_t state = front[fi++];
        const edges_t* es = &trans[state];
        for (uint32_t k = es->n; k > 0; k--) rev[nr++] = (uint32_t)es->e[k - 1].next;   /* extra ++ revBacklog */
        order[n_order++] = state;
    }
    free(front); free(rev);
    return order;
}

/* Automaton.hs:176-200 build, with
 *   :249-292 buildTransitionMap, :336-362 buildFallbackMap, :367-380 buildValueMap,
 *   :190-192 makeTransitions, :166-172 packTransitions, :301-306 root ASCII table.
 * Needle i carries value `values_in ? values_in[i] : i`. */
orc_machine* orc_build(const uint8_t* bytes, const uint64_t* offs, size_t n_needles, const uint32_t* values_in)
{
    size_t cap_states = 1024, n_states = 1;
    edges_t* trans = (edges_t*)calloc(cap_states, sizeof(edges_t));
    vals_t* initial = (vals_t*)calloc(cap_states, sizeof(vals_t));

    for (size_t i = 0; i < n_needles; i++) {                       /* List.foldl' insertNeedle (:292) */
        const uint8_t* nd = bytes + offs[i];
        size_t nlen = (size_t)(offs[i + 1] - offs[i]);
        uint32_t value = values_in ? values_in[i] : (uint32_t)i;
        uint32_t state = 0; size_t index = 0;
        while (index < nlen) {                                     /* go (:258-284) */
            size_t units; uint32_t cp = decode_at(nd, index, nlen, &units);
            int k = edges_find(&trans[state], (int32_t)cp);
            if (k >= 0) {
                state = (uint32_t)trans[state].e[k].next;
            } else {
                if (n_states == cap_states) {
                    trans = (edges_t*)realloc(trans, cap_states * 2 * sizeof(edges_t));
                    initial = (vals_t*)realloc(initial, cap_states * 2 * sizeof(vals_t));
                    memset(trans + cap_states, 0, cap_states * sizeof(edges_t));
                    memset(initial + cap_states, 0, cap_states * sizeof(vals_t));
                    cap_states *= 2;
                }
                uint32_t next = (uint32_t)n_states++;              /* nextState = numStates (:278) */
                edges_insert(&trans[state], (int32_t)cp, (int32_t)next);
                state = next;
            }
            index += units;
        }
        /* IntMap.insertWith (++) state [value] values (:263): new value goes in FRONT */
        vals_t* vs = &initial[state];
        if (vs->n == vs->cap) { vs->cap = vs->cap ? vs->cap * 2 : 1; vs->v = (uint32_t*)realloc(vs->v, vs->cap * sizeof(uint32_t)); }
        memmove(vs->v + 1, vs->v, vs->n * sizeof(uint32_t));
        vs->v[0] = value; vs->n++;
    }

    uint32_t* order = bfs_order(trans, n_states);

    /* buildFallbackMap (:336-362) */
    uint32_t* fallback = (uint32_t*)calloc(n_states, sizeof(uint32_t));
    for (size_t oi = 0; oi < n_states; oi++) {
        uint32_t state = order[oi];
        const edges_t* es = &trans[state];
        for (uint32_t k = 0; k < es->n; k++) {
            int32_t input = es->e[k].cp; uint32_t next = (uint32_t)es->e[k].next;
            uint32_t st = state, fb = 0;
            for (;;) {                                             /* getFallback (:342-352) */
                if (st == 0) { fb = 0; break; }
                uint32_t f = fallback[st];
                int j = edges_find(&trans[f], input);
                if (j >= 0) { fb = (uint32_t)trans[f].e[j].next; break; }
                st = f;
            }
            fallback[next] = fb;
        }
    }

    /* buildValueMap (:367-380): values[s] = initial[s] ++ values[fallback[s]] in BFS order */
    vals_t* values = (vals_t*)calloc(n_states, sizeof(vals_t));
    for (size_t oi = 0; oi < n_states; oi++) {
        uint32_t state = order[oi];
        const vals_t* fbv = &values[fallback[state]];              /* root: values ! 0 is still [] here */
        uint32_t fbn = (state == 0) ? 0 : fbv->n;
        uint32_t n = initial[state].n + fbn;
        if (n) {
            values[state].v = (uint32_t*)malloc(n * sizeof(uint32_t));
            if (initial[state].n) memcpy(values[state].v, initial[state].v, initial[state].n * sizeof(uint32_t));      /* (memcpy of zero bytes from NULL is undefined: UBSan, round 6) */
            if (fbn) memcpy(values[state].v + initial[state].n, fbv->v, fbn * sizeof(uint32_t));
        }
        values[state].n = n;
    }

    orc_machine* m = (orc_machine*)calloc(1, sizeof(orc_machine));
    m->n_states = n_states;
    m->fallback = fallback;

    /* makeTransitions (:190-192) + packTransitions (:166-172): per state the goto edges in
     * DESCENDING code point order (ascending fold that prepends), then the wildcard. */
    size_t total = 0;
    for (size_t s = 0; s < n_states; s++) total += trans[s].n + 1;
    m->n_transitions = total;
    m->transitions = (uint64_t*)malloc(total * sizeof(uint64_t));
    m->offsets = (uint32_t*)malloc((n_states + 1) * sizeof(uint32_t));
    size_t w = 0;
    for (size_t s = 0; s < n_states; s++) {
        m->offsets[s] = (uint32_t)w;
        for (uint32_t k = trans[s].n; k > 0; k--)
            m->transitions[w++] = new_transition((uint32_t)trans[s].e[k - 1].cp, (uint32_t)trans[s].e[k - 1].next);
        m->transitions[w++] = new_wildcard(fallback[s]);
    }
    m->offsets[n_states] = (uint32_t)w;

    /* buildAsciiTransitionLookupTable (:301-306) */
    for (uint32_t i = 0; i < 128; i++) {
        int k = edges_find(&trans[0], (int32_t)i);
        m->root_ascii[i] = k >= 0 ? new_transition(i, (uint32_t)trans[0].e[k].next) : new_wildcard(0);
    }

    /* Vector.generate numStates (valueMap !) (:198) */
    m->values_off = (uint64_t*)malloc((n_states + 1) * sizeof(uint64_t));
    uint64_t tv = 0;
    for (size_t s = 0; s < n_states; s++) { m->values_off[s] = tv; tv += values[s].n; }
    m->values_off[n_states] = tv;
    m->values = (uint32_t*)malloc((tv ? tv : 1) * sizeof(uint32_t));
    for (size_t s = 0; s < n_states; s++)
        if (values[s].n) memcpy(m->values + m->values_off[s], values[s].v, values[s].n * sizeof(uint32_t));

    for (size_t s = 0; s < n_states; s++) { free(trans[s].e); free(initial[s].v); free(values[s].v); }
    free(trans); free(initial); free(values); free(order);
    return m;
}

void orc_free(orc_machine* m)
{
    if (!m) return;
    free(m->transitions); free(m->offsets); free(m->values); free(m->values_off); free(m->fallback); free(m);
}

size_t orc_num_states(const orc_machine* m) { return m->n_states; }
size_t orc_num_transitions(const orc_machine* m) { return m->n_transitions; }
const uint64_t* orc_transitions(const orc_machine* m) { return m->transitions; }
const uint32_t* orc_offsets(const orc_machine* m) { return m->offsets; }
const uint64_t* orc_root_ascii(const orc_machine* m) { return m->root_ascii; }
const uint64_t* orc_values_off(const orc_machine* m) { return m->values_off; }
const uint32_t* orc_values(const orc_machine* m) { return m->values; }
const uint32_t* orc_fallback(const orc_machine* m) { return m->fallback; }

/* ------------------------------------------------------------------ running the machine */

/* Automaton.hs:398  data Next a = Done !a | Step !a.  The fold function mutates *acc and
 * returns ORC_DONE or ORC_STEP. */
#define ORC_STEP 0
#define ORC_DONE 1
typedef int (*orc_fold_fn)(void* acc, uint64_t match_pos, uint32_t match_value);

/* Automaton.hs:442-534 runWithCase.  The join points of the reference are the labels
 * below; `d + off` is the start of the Text slice, matchPos is relative to it (:452,:530). */
void orc_run_with_case(int ignore_case, void* acc, orc_fold_fn f, const orc_machine* m,
                       const uint8_t* d, size_t off, size_t len)
{
    const uint64_t* transitions = m->transitions;
    const uint32_t* offsets = m->offsets;
    const size_t initial_offset = off, limit = off + len;
    size_t offset = initial_offset;
    uint32_t state = 0, cp;
    uint64_t t; uint32_t i;

consume_input:                                   /* :468-480 */
    if (offset >= limit) return;
    {
        size_t units;
        cp = decode_at(d, offset, limit, &units);
        if (ignore_case) cp = orc_lower_code_point(cp);
        offset += units;
    }
follow_code_point:                               /* :482-486 */
    if (state == 0 && cp < 128) {                /* lookupRootAsciiTransition :514-520 */
        t = m->root_ascii[cp];
        if (t_is_wildcard(t)) { state = 0; goto consume_input; }
        state = t_state(t);
        goto collect_matches;
    }
    i = offsets[state];
lookup_transition:                               /* :489-510 */
    t = transitions[i];
    if (t_is_wildcard(t)) {
        if (state == 0) goto consume_input;
        state = t_state(t);
        goto follow_code_point;
    }
    if (t_code(t) == cp) { state = t_state(t); goto collect_matches; }
    i++;
    goto lookup_transition;
collect_matches:                                 /* :522-534 */
    {
        const uint32_t* vs = m->values + m->values_off[state];
        uint64_t nv = m->values_off[state + 1] - m->values_off[state];
        for (uint64_t k = 0; k < nv; k++)
            if (f(acc, (uint64_t)(offset - initial_offset), vs[k]) == ORC_DONE) return;
    }
    goto consume_input;
}

/* tests/Data/Text/AhoCorasickSpec.hs:252-261 and benchmark/haskell/app/Main.hs:67-76 countMatches */
static int fold_count(void* acc, uint64_t pos, uint32_t v) { (void)pos; (void)v; (*(uint64_t*)acc)++; return ORC_STEP; }
uint64_t orc_count_matches(const orc_machine* m, int ignore_case, const uint8_t* d, size_t off, size_t len)
{
    uint64_t n = 0;
    orc_run_with_case(ignore_case, &n, fold_count, m, d, off, len);
    return n;
}

/* All matches in fold order (oldest first; the README prints the list newest-first). */
typedef struct { uint64_t* pos; uint32_t* val; uint64_t n, cap; } list_acc;
static int fold_list(void* a, uint64_t pos, uint32_t v)
{
    list_acc* l = (list_acc*)a;
    if (l->n < l->cap) { l->pos[l->n] = pos; l->val[l->n] = v; }
    l->n++;
    return ORC_STEP;
}
uint64_t orc_run_list(const orc_machine* m, int ignore_case, const uint8_t* d, size_t off, size_t len,
                      uint64_t* pos_out, uint32_t* val_out, uint64_t cap)
{
    list_acc l = { pos_out, val_out, 0, cap };
    orc_run_with_case(ignore_case, &l, fold_list, m, d, off, len);
    return l.n;
}

/* Checksum of the whole fold sequence (SURVEY 8d "parity check at scale"): runWithCase with the fold
 *   f (h, n) (Match pos v) = Step (h * P + mix pos v, n + 1)
 * order-sensitive in (pos, v), so two runs agree iff they hand the same matches to the fold in the same order
 * (up to 64-bit collisions).  libam computes the same fold over its records on the device (am_matches_fold_hash). */
#define ORC_HASH_P 0x100000001B3ull
static inline uint64_t orc_mix(uint64_t pos, uint32_t v)
{
    uint64_t x = (pos * 0x9E3779B97F4A7C15ull) ^ ((uint64_t)v + 0x632BE59BD9B4E019ull);
    x ^= x >> 32; x *= 0xD6E8FEB86659FD93ull; x ^= x >> 32;
    return x;
}
typedef struct { uint64_t h, n; } hash_acc;
static int fold_hash(void* a, uint64_t pos, uint32_t v)
{
    hash_acc* h = (hash_acc*)a;
    h->h = h->h * ORC_HASH_P + orc_mix(pos, v);
    h->n++;
    return ORC_STEP;
}
uint64_t orc_fold_hash(const orc_machine* m, int ignore_case, const uint8_t* d, size_t off, size_t len, uint64_t* count_out)
{
    hash_acc h = { 0, 0 };
    orc_run_with_case(ignore_case, &h, fold_hash, m, d, off, len);
    if (count_out) *count_out = h.n;
    return h.h;
}

/* src/Data/Text/AhoCorasick/Searcher.hs:156-164 containsAny */
static int fold_any(void* acc, uint64_t pos, uint32_t v) { (void)pos; (void)v; *(int*)acc = 1; return ORC_DONE; }
int orc_contains_any(const orc_machine* m, int ignore_case, const uint8_t* d, size_t off, size_t len)
{
    int found = 0;
    orc_run_with_case(ignore_case, &found, fold_any, m, d, off, len);
    return found;
}

/* src/Data/Text/AhoCorasick/Searcher.hs:173-187 containsAll (machine built with values = needle ids
 * 0..n-1, :167-169).  IntSet restated as a bitmap + remaining counter. */
typedef struct { uint8_t* present; size_t remaining; } all_acc;
static int fold_all(void* a, uint64_t pos, uint32_t needle_id)
{
    all_acc* s = (all_acc*)a; (void)pos;
    if (s->present[needle_id]) { s->present[needle_id] = 0; s->remaining--; }
    return s->remaining == 0 ? ORC_DONE : ORC_STEP;
}
int orc_contains_all(const orc_machine* m, int ignore_case, size_t n_needles, const uint8_t* d, size_t off, size_t len)
{
    all_acc s; s.present = (uint8_t*)malloc(n_needles ? n_needles : 1); s.remaining = n_needles;
    memset(s.present, 1, n_needles);
    orc_run_with_case(ignore_case, &s, fold_all, m, d, off, len);
    free(s.present);
    return s.remaining == 0;
}

/* ------------------------------------------------------------------ Replacer */

/* Replacer.hs:59-70 Payload */
typedef struct { int64_t priority; size_t len_bytes; size_t len_cps; const uint8_t* repl; size_t repl_len; } payload_t;
/* Replacer.hs:159 Match pos len replacement, Ord derived */
typedef struct { int64_t pos; int64_t len; const uint8_t* repl; size_t repl_len; } rmatch_t;

typedef struct orc_replacer {
    orc_machine* machine;
    payload_t* payloads; size_t n_needles;
    uint8_t* repl_store;
    int ignore_case;
} orc_replacer;

/* Replacer.hs:97-116 build.  IgnoreCase lower-cases the needle (:105-107); the payload lengths
 * come from the ORIGINAL needle (:112-113). */
orc_replacer* orc_replacer_build(int ignore_case, const uint8_t* nbytes, const uint64_t* noffs,
                                 const uint8_t* rbytes, const uint64_t* roffs, size_t n)
{
    orc_replacer* r = (orc_replacer*)calloc(1, sizeof(orc_replacer));
    r->ignore_case = ignore_case; r->n_needles = n;
    r->payloads = (payload_t*)calloc(n ? n : 1, sizeof(payload_t));
    size_t rtotal = (size_t)roffs[n];
    r->repl_store = (uint8_t*)malloc(rtotal ? rtotal : 1);
    memcpy(r->repl_store, rbytes, rtotal);

    uint64_t* offs2 = (uint64_t*)malloc((n + 1) * sizeof(uint64_t));
    size_t cap = 16, used = 0; uint8_t* buf = (uint8_t*)malloc(cap);
    for (size_t i = 0; i < n; i++) {
        const uint8_t* nd = nbytes + noffs[i]; size_t nlen = (size_t)(noffs[i + 1] - noffs[i]);
        r->payloads[i].priority = -(int64_t)i;
        r->payloads[i].len_bytes = nlen;
        r->payloads[i].len_cps = count_code_points(nd, nlen);
        r->payloads[i].repl = r->repl_store + roffs[i];
        r->payloads[i].repl_len = (size_t)(roffs[i + 1] - roffs[i]);
        uint8_t* low = NULL; size_t low_len = nlen; const uint8_t* src = nd;
        if (ignore_case) { low = orc_lower_utf8(nd, nlen, &low_len); src = low; }
        while (used + low_len > cap) { cap *= 2; buf = (uint8_t*)realloc(buf, cap); }
        offs2[i] = used; memcpy(buf + used, src, low_len); used += low_len;
        free(low);
    }
    offs2[n] = used;
    r->machine = orc_build(buf, offs2, n, NULL);     /* value = needle index -> payloads[] */
    free(buf); free(offs2);
    return r;
}

/* Replacer.hs:148-153 setCaseSensitivity: the needles in the automaton stay as they are (an IgnoreCase-built replacer keeps its
 * lower-cased needles), the payloads keep the ORIGINAL needles' lengths; only the mode of the scan and of makeMatch (:264-274) changes. */
void orc_replacer_set_case(orc_replacer* r, int ignore_case) { r->ignore_case = ignore_case; }

void orc_replacer_free(orc_replacer* r)
{
    if (!r) return;
    orc_free(r->machine); free(r->payloads); free(r->repl_store); free(r);
}

static int cmp_bytes(const uint8_t* a, size_t na, const uint8_t* b, size_t nb)
{
    size_t n = na < nb ? na : nb;
    int c = n ? memcmp(a, b, n) : 0;
    if (c) return c;
    return (na > nb) - (na < nb);
}
/* derived Ord on Match (Replacer.hs:159): pos, then len, then replacement Text
 * (Text's Ord is code point order = bytewise order on UTF-8). */
static int cmp_rmatch(const void* pa, const void* pb)
{
    const rmatch_t* a = (const rmatch_t*)pa; const rmatch_t* b = (const rmatch_t*)pb;
    if (a->pos != b->pos) return a->pos < b->pos ? -1 : 1;
    if (a->len != b->len) return a->len < b->len ? -1 : 1;
    return cmp_bytes(a->repl, a->repl_len, b->repl, b->repl_len);
}

typedef struct {
    const orc_replacer* r; int64_t threshold; int64_t p_best;
    const uint8_t* hay; size_t hay_len;
    rmatch_t* ms; size_t n, cap;
} prep_acc;

/* Replacer.hs:252-274 prependMatch + makeMatch */
static int fold_prepend(void* a, uint64_t pos, uint32_t value)
{
    prep_acc* s = (prep_acc*)a;
    const payload_t* p = &s->r->payloads[value];
    if (!(p->priority < s->threshold && p->priority >= s->p_best)) return ORC_STEP;
    if (p->priority > s->p_best) { s->p_best = p->priority; s->n = 0; }
    rmatch_t mt;
    if (!s->r->ignore_case) { mt.pos = (int64_t)pos - (int64_t)p->len_bytes; mt.len = (int64_t)p->len_bytes; }
    else {
        int64_t start = orc_skip_code_points_backwards(s->hay, s->hay_len, (int64_t)pos - 1, (int64_t)p->len_cps - 1);
        mt.pos = start; mt.len = (int64_t)pos - start;
    }
    mt.repl = p->repl; mt.repl_len = p->repl_len;
    if (s->n == s->cap) { s->cap = s->cap ? s->cap * 2 : 16; s->ms = (rmatch_t*)realloc(s->ms, s->cap * sizeof(rmatch_t)); }
    s->ms[s->n++] = mt;       /* list order is irrelevant: the caller sorts (:241-242) */
    return ORC_STEP;
}

/* Replacer.hs:203-242 runWithLimit.  Returns malloc'd bytes, or NULL for Nothing
 * (result longer than max_len).  max_len < 0 means maxBound (Replacer.hs:200-201 run). */
uint8_t* orc_replacer_run(const orc_replacer* r, const uint8_t* hay, size_t hay_len, int64_t max_len, size_t* out_len)
{
    uint8_t* cur = (uint8_t*)malloc(hay_len ? hay_len : 1);
    memcpy(cur, hay, hay_len);
    size_t cur_len = hay_len;
    int64_t threshold = 1;                                   /* initialThreshold :211 */
    int64_t min_priority = 1 - (int64_t)r->n_needles;        /* :217 */
    prep_acc s; memset(&s, 0, sizeof(s)); s.r = r;
    for (;;) {
        s.threshold = threshold; s.p_best = INT64_MIN; s.n = 0; s.hay = cur; s.hay_len = cur_len;   /* seed :222 */
        orc_run_with_case(r->ignore_case, &s, fold_prepend, r->machine, cur, 0, cur_len);
        if (s.n == 0) break;                                 /* (_, []) -> Just haystack :230 */
        /* replacementLength (:183-187) over ALL matches, before overlap removal (:240) */
        int64_t newlen = (int64_t)cur_len;
        for (size_t k = 0; k < s.n; k++) newlen += (int64_t)s.ms[k].repl_len - s.ms[k].len;
        if (max_len >= 0 && newlen > max_len) { free(cur); free(s.ms); return NULL; }
        qsort(s.ms, s.n, sizeof(rmatch_t), cmp_rmatch);      /* sort :241-242 */
        /* removeOverlap (:191-198) */
        size_t w = 0;
        for (size_t k = 0; k < s.n; k++) {
            if (w == 0 || s.ms[k].pos >= s.ms[w - 1].pos + s.ms[w - 1].len) s.ms[w++] = s.ms[k];
        }
        /* replace (:163-180) */
        size_t outcap = cur_len + 1;
        for (size_t k = 0; k < w; k++) outcap += s.ms[k].repl_len;
        uint8_t* out = (uint8_t*)malloc(outcap);
        size_t o = 0; int64_t at = 0;
        for (size_t k = 0; k < w; k++) {
            memcpy(out + o, cur + at, (size_t)(s.ms[k].pos - at)); o += (size_t)(s.ms[k].pos - at);
            memcpy(out + o, s.ms[k].repl, s.ms[k].repl_len); o += s.ms[k].repl_len;
            at = s.ms[k].pos + s.ms[k].len;
        }
        memcpy(out + o, cur + at, cur_len - (size_t)at); o += cur_len - (size_t)at;
        free(cur); cur = out; cur_len = o;
        if (s.p_best == min_priority) break;                 /* :241 */
        threshold = s.p_best;                                /* go p newHaystack :242 */
    }
    free(s.ms);
    *out_len = cur_len;
    return cur;
}

void orc_free_bytes(uint8_t* p) { free(p); }
