/*
 * oracle/n2v_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * CPU restatement of the node2vec path GEM runs through the prebuilt SNAP binary
 * gem/c_exe/node2vec (call site gem/embedding/node2vec.py:34-48).  As GEM runs it the
 * binary seeds its RNG with time() and trains Hogwild over OpenMP threads, so it has no
 * reproducible output.  PINNED at the vector level all the same, through
 * oracle/snap_stream.py: with time() fixed (oracle/shim/faketime.c) and one thread the
 * binary is deterministic, and snap_stream.py -- the semantics of this file on the
 * binary's own sequential TRnd stream, in its fp64 -- reproduces (i) walk matrices dumped
 * from the running binary bit for bit and (ii) the embedding files those runs wrote to
 * the six digits the binary prints (tests/golden/n2v_snap_stream_walks.json).  THIS file
 * is tied to that restatement in tests/test_oracle_n2v.py: the same walk body and the
 * same TrainModel body, fed with this file's counter-based draws, give this file's walks
 * bit for bit and this file's embeddings to fp32 rounding (2e-5); alias targets equal;
 * second-order frequencies follow the pinned per-pair tables.  What stays statistical is
 * only what cannot be otherwise: Hogwild launches, and p,q != 1 walks (rejection here,
 * per-pair tables there: same distribution, different draws).  The source is third-party
 * (snap-stanford/snap, examples/node2vec + snap-adv/{n2v,biasedrandomwalk,word2vec}.cpp,
 * not vendored, not version pinned -- gem/c_exe/readme.txt:1; ELF banner "Apr 9 2017").
 * This file restates the published algorithm; constants and the sampling quirks were
 * checked against the ELF's symbols/disassembly:
 *   GetNodeAlias      @0x4115f0  Vose alias build, stacks popped from the back
 *   AliasDrawInt      @0x411360  X=floor(u*N); Y<prob[X] ? X : alias[X]
 *   SimulateWalk      @0x411a00  first hop UNIFORM, then alias draws; stops at sinks;
 *                                walks live in a zero-initialised matrix (short walks
 *                                leave trailing 0 tokens, trained as node 0)
 *   PreprocessNode    @0x411f40  2nd-order weights w/p (x==t), w (x in N(t)), w/q (else)
 *   LearnVocab        @0x40d560  token counts
 *   InitUnigramTable  @0x40e520  count^0.75, Vose in fp64
 *   RndUnigramInt     @0x40d5f0  X = KTable[floor(u*n)]  (sic: the alias of a random
 *                                slot, not the slot) ; Y<UTable[X] ? X : KTable[X]
 *   InitPosEmb        @0x40e270  (U(0,1)-0.5)/d ;  InitNegEmb @0x40e040 zeros
 *   TrainModel        @0x40d6a0  StartAlpha .025, floor .025e-4, alpha refreshed every
 *                                10000 words, window shrink b=rand%k, 1+5 targets,
 *                                sigmoid clamp +-6 (table in the ELF; exact here)
 * Statistical parity against the real binary is tested in tests/ (MAP of >=3 SNAP runs
 * in tests/golden/n2v_ref.json).
 *
 * Randomness: the reference's TRnd stream is not reproducible, so oracle and device
 * share a COUNTER-BASED stream (Philox4x32-10, restated below): every draw is a pure
 * function of (seed, walk, step/position, purpose).  Integer outputs (walks, counts,
 * alias tables) must match the device bit for bit; embeddings to fp32 tolerance.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------ Philox4x32-10 */
typedef struct { uint32_t x, y, z, w; } u32x4;

static u32x4 philox(uint64_t seed, uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3)
{
    uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
    for (int r = 0; r < 10; ++r) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c0;
        const uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n1 = (uint32_t)p1;
        const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1, n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    u32x4 o = {c0, c1, c2, c3};
    return o;
}
static float u01(uint32_t r) { return (float)(r >> 8) * (1.0f / 16777216.0f); }
/* floor(u * n) for u = r / 2^32, exact in integers */
static uint32_t mulhi_range(uint32_t r, uint32_t n) { return (uint32_t)(((uint64_t)r * n) >> 32); }

/* purposes (counter word c3, low byte) */
enum { TAG_WALK = 1, TAG_WIN = 2, TAG_NEG = 3, TAG_INIT = 4 };

void oracle_philox(uint64_t seed, uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t *out)
{
    u32x4 r = philox(seed, c0, c1, c2, c3);
    out[0] = r.x; out[1] = r.y; out[2] = r.z; out[3] = r.w;
}

/* ---------------------------------------------- start-node permutation (shuffle)
 * SNAP shuffles the node list before every walk round (NIdsV.Shuffle).  Stateless
 * stand-in: a 4-round Feistel bijection on 2*hb bits, cycle-walked into [0, n). */
static uint32_t fmix32(uint32_t h)
{
    h ^= h >> 16; h *= 0x85EBCA6Bu; h ^= h >> 13; h *= 0xC2B2AE35u; h ^= h >> 16;
    return h;
}
uint32_t oracle_perm(uint32_t j, uint32_t n, uint64_t key)
{
    uint32_t bits = 1;
    while (((uint64_t)1 << bits) < n) ++bits;
    const uint32_t hb = (bits + 1) / 2, mask = (1u << hb) - 1u;
    do {
        uint32_t L = j >> hb, R = j & mask;
        for (uint32_t r = 0; r < 4; ++r) {
            const uint32_t k = (uint32_t)(key >> ((r & 1u) * 32)) + r * 0x9E3779B9u;
            const uint32_t t = L ^ (fmix32(R + k) & mask);
            L = R; R = t;
        }
        j = (L << hb) | R;
    } while (j >= n);
    return j;
}

/* ----------------------------------------------------------- Vose alias (fp32)
 * GetNodeAlias: P normalised to sum 1; U[i]=P[i]*N; small/large stacks filled in
 * index order and popped from the BACK; leftovers get U=1.  K initialised to 0.
 * `work` is scratch of N ints (small stack grows up from 0, large stack down from N-1). */
void oracle_alias_build_f32(int32_t N, const float *wts, float *U, int32_t *K, int32_t *work)
{
    float sum = 0.0f;
    for (int32_t i = 0; i < N; ++i) sum += wts[i];
    int32_t ns = 0, nl = 0;
    for (int32_t i = 0; i < N; ++i) {
        K[i] = 0;
        U[i] = (wts[i] / sum) * (float)N;
        if (U[i] < 1.0f) work[ns++] = i; else work[N - 1 - nl++] = i;
    }
    while (ns > 0 && nl > 0) {
        const int32_t s = work[--ns];
        const int32_t l = work[N - nl]; --nl;
        K[s] = l;
        U[l] = U[l] + U[s] - 1.0f;
        if (U[l] < 1.0f) work[ns++] = l; else work[N - 1 - nl++] = l;
    }
    while (ns > 0) U[work[--ns]] = 1.0f;
    while (nl > 0) { U[work[N - nl]] = 1.0f; --nl; }
}

/* ------------------------------------------------- Vose alias for HUB rows (N >= ORACLE_ALIAS_HUB_DEG)
 * The same table GetNodeAlias builds, without its N-step sequential loop.  Its two stacks are filled in index order and popped from the back, so
 * the smalls are met in DESCENDING index order s_1, s_2, ... and so are the larges L_1, L_2, ...; a large stays on top of its stack until it has
 * absorbed more than its excess, then turns small and is absorbed at once by the NEXT large.  With D_j = sum_{i<=j} (1 - u[s_i]) (deficits met so
 * far) and E_m = sum_{i<=m} (u[L_i] - 1) (excess spent so far) the loop's outcome is closed form:
 *   small s_j goes to  K = L_m,  m = min{m : E_m >= D_{j-1}}               (none: the larges ran out, U = 1)
 *   large L_m turns small when  D_j > E_m  first holds: U = 1 + E_m - D_j, K = L_{m+1}   (never, or no L_{m+1}: U = 1)
 * i.e. two prefix sums and two binary searches per entry -- what gem_amd/csrc/n2v.hip's n2v_alias_hub_kernel runs with a workgroup per row (a
 * 94 115-neighbour R-MAT hub is a 94k-step chain on ONE lane otherwise).  fp64, and the ORDER of every sum is fixed -- chunks of 256 entries in
 * stack order, sequential inside a chunk, chunk totals accumulated sequentially, entry = chunk base + running sum inside the chunk -- so that this
 * file and the kernel round identically: tables bit for bit.  Equal to the sequential loop in fp64 (tests/test_oracle_n2v.py: same K as the
 * restatement pinned to the binary, U to 1e-12). */
#define ORACLE_ALIAS_HUB_DEG 2048
#define ORACLE_ALIAS_CHUNK 256
void oracle_alias_build_hub(int32_t N, const float *wts, float *U, int32_t *K)
{
    const int32_t nch = (N + ORACLE_ALIAS_CHUNK - 1) / ORACLE_ALIAS_CHUNK;
    double *cd = (double *)malloc(sizeof(double) * (size_t)nch * 2), *ce = cd + nch;
    int32_t *cs = (int32_t *)malloc(sizeof(int32_t) * (size_t)nch * 2), *cl = cs + nch;
    int32_t *S = (int32_t *)malloc(sizeof(int32_t) * (size_t)N * 2), *L = S + N;
    double *D = (double *)malloc(sizeof(double) * (size_t)N * 2), *E = D + N;
    double total = 0.0;
    for (int32_t c = 0; c < nch; ++c) {                       /* chunks of the INDEX order for the total */
        double t = 0.0;
        const int32_t i1 = (c + 1) * ORACLE_ALIAS_CHUNK < N ? (c + 1) * ORACLE_ALIAS_CHUNK : N;
        for (int32_t i = c * ORACLE_ALIAS_CHUNK; i < i1; ++i) t += (double)wts[i];
        total += t;
    }
    /* stack order: r = 0 is the top of both stacks, i = N - 1 - r */
    for (int32_t c = 0; c < nch; ++c) {
        double ds = 0.0, es = 0.0; int32_t ns = 0, nl = 0;
        const int32_t r1 = (c + 1) * ORACLE_ALIAS_CHUNK < N ? (c + 1) * ORACLE_ALIAS_CHUNK : N;
        for (int32_t r = c * ORACLE_ALIAS_CHUNK; r < r1; ++r) {
            const double u = ((double)wts[N - 1 - r] / total) * (double)N;
            if (u < 1.0) { ds += 1.0 - u; ++ns; } else { es += u - 1.0; ++nl; }
        }
        cd[c] = ds; ce[c] = es; cs[c] = ns; cl[c] = nl;
    }
    double bd = 0.0, be = 0.0; int32_t bs = 0, bl = 0;        /* exclusive prefix over the chunks, sequential */
    for (int32_t c = 0; c < nch; ++c) {
        const double ds = cd[c], es = ce[c]; const int32_t ns = cs[c], nl = cl[c];
        cd[c] = bd; ce[c] = be; cs[c] = bs; cl[c] = bl;
        bd += ds; be += es; bs += ns; bl += nl;
    }
    const int32_t NS = bs, NL = bl;
    for (int32_t c = 0; c < nch; ++c) {
        double ds = 0.0, es = 0.0; int32_t ns = cs[c], nl = cl[c];
        const int32_t r1 = (c + 1) * ORACLE_ALIAS_CHUNK < N ? (c + 1) * ORACLE_ALIAS_CHUNK : N;
        for (int32_t r = c * ORACLE_ALIAS_CHUNK; r < r1; ++r) {
            const int32_t i = N - 1 - r;
            const double u = ((double)wts[i] / total) * (double)N;
            K[i] = 0;
            if (u < 1.0) { ds += 1.0 - u; S[ns] = i; D[ns] = cd[c] + ds; ++ns; U[i] = (float)u; }
            else { es += u - 1.0; L[nl] = i; E[nl] = ce[c] + es; ++nl; U[i] = 1.0f; }
        }
    }
    for (int32_t j = 0; j < NS; ++j) {                        /* smalls: first m with E[m] >= D[j-1] */
        const double dp = j ? D[j - 1] : 0.0;
        int32_t lo = 0, hi = NL;
        while (lo < hi) { const int32_t mid = (lo + hi) >> 1; if (E[mid] >= dp) hi = mid; else lo = mid + 1; }
        if (lo < NL) K[S[j]] = L[lo]; else U[S[j]] = 1.0f;
    }
    for (int32_t m = 0; m < NL; ++m) {                        /* larges: first j with D[j] > E[m] */
        int32_t lo = 0, hi = NS;
        while (lo < hi) { const int32_t mid = (lo + hi) >> 1; if (D[mid] > E[m]) hi = mid; else lo = mid + 1; }
        if (lo < NS && m + 1 < NL) { U[L[m]] = (float)(1.0 + E[m] - D[lo]); K[L[m]] = L[m + 1]; }
    }
    free(cd); free(cs); free(S); free(D);
}

/* per-row first-order tables over a CSR graph */
void oracle_n2v_alias_rows(int64_t n, const int64_t *row_ptr, const float *w, float *U, int32_t *K)
{
    int64_t maxdeg = 1;
    for (int64_t v = 0; v < n; ++v)
        if (row_ptr[v + 1] - row_ptr[v] > maxdeg) maxdeg = row_ptr[v + 1] - row_ptr[v];
    int32_t *work = (int32_t *)malloc(sizeof(int32_t) * (size_t)maxdeg);
    for (int64_t v = 0; v < n; ++v) {
        const int64_t a = row_ptr[v], deg = row_ptr[v + 1] - a;
        if (deg >= ORACLE_ALIAS_HUB_DEG) oracle_alias_build_hub((int32_t)deg, w + a, U + a, K + a);
        else if (deg > 0) oracle_alias_build_f32((int32_t)deg, w + a, U + a, K + a, work);
    }
    free(work);
}

/* ------------------------------------------------------------------- walks
 * CSR with columns SORTED inside each row (needed for the has_edge(t,x) test).
 * U/K = first-order alias tables (NULL => all weights in a row equal => uniform pick).
 * Second order (p,q != 1): rejection sampling -- propose x ~ w(v,.), accept with
 * a(t,x)/amax, a = 1/p if x==t, 1 if (t->x) is an edge, 1/q otherwise.  Same
 * distribution as SNAP's per-(t,v) alias tables (PreprocessNode) without their
 * sum-of-squared-degrees memory.
 * flags bit0: pad short walks with 0 (SNAP) else -1; bit3: uniform first hop (SNAP). */
static int has_edge_sorted(const int64_t *row_ptr, const int32_t *col, int32_t t, int32_t x)
{
    int64_t lo = row_ptr[t], hi = row_ptr[t + 1];
    while (lo < hi) {
        const int64_t mid = (lo + hi) >> 1;
        if (col[mid] < x) lo = mid + 1; else hi = mid;
    }
    return lo < row_ptr[t + 1] && col[lo] == x;
}

void oracle_n2v_walks(int64_t n, const int64_t *row_ptr, const int32_t *col, const float *U, const int32_t *K,
                      float p, float q, int32_t num_walks, int32_t walk_len, uint64_t seed, int32_t flags,
                      int64_t walk_begin, int64_t walk_end, const int32_t *start, int64_t m,
                      int32_t *walks /* [(walk_end-walk_begin)][walk_len] */)
{
    /* start[0..m): the nodes that occur in the edge list, ascending -- the binary builds its graph from the edge
     * list, so an isolated node never starts a walk (node2vec(): `for NI = InNet->BegNI()...`). */
    (void)num_walks; (void)n;
    const int32_t pad = (flags & 1) ? 0 : -1;
    const int second = !(p == 1.0f && q == 1.0f);
    const float ip = 1.0f / p, iq = 1.0f / q;
    float amax = 1.0f;
    if (ip > amax) amax = ip;
    if (iq > amax) amax = iq;
    for (int64_t wid = walk_begin; wid < walk_end; ++wid) {
        int32_t *out = walks + (wid - walk_begin) * walk_len;
        const uint32_t round = (uint32_t)(wid / m), j = (uint32_t)(wid % m);
        int32_t cur = start[oracle_perm(j, (uint32_t)m, seed ^ ((uint64_t)(round + 1) * 0x9E3779B97F4A7C15ull))];
        int32_t prev = -1;
        int32_t len = 0;
        out[len++] = cur;
        while (len < walk_len) {
            const int64_t a = row_ptr[cur];
            const uint32_t deg = (uint32_t)(row_ptr[cur + 1] - a);
            if (deg == 0) break;
            int32_t nxt = -1;
            for (uint32_t trial = 0;; ++trial) {
                const u32x4 r = philox(seed, (uint32_t)wid, (uint32_t)((uint64_t)wid >> 32), (uint32_t)len, TAG_WALK | (trial << 8));
                uint32_t slot = mulhi_range(r.x, deg);
                if (U && !(len == 1 && (flags & 8))) {
                    if (!(u01(r.y) < U[a + slot])) slot = (uint32_t)K[a + slot];
                }
                const int32_t x = col[a + slot];
                if (!second || len == 1) { nxt = x; break; }
                const float alpha = (x == prev) ? ip : (has_edge_sorted(row_ptr, col, prev, x) ? 1.0f : iq);
                if (u01(r.z) * amax < alpha || trial >= 4095) { nxt = x; break; }
            }
            prev = cur; cur = nxt;
            out[len++] = cur;
        }
        for (; len < walk_len; ++len) out[len] = pad;
    }
}

/* LearnVocab: token counts (pad tokens -1 are skipped; SNAP's pad 0 counts as node 0) */
void oracle_n2v_vocab(int64_t n, int64_t ntokens, const int32_t *walks, int32_t *counts)
{
    memset(counts, 0, sizeof(int32_t) * (size_t)n);
    for (int64_t i = 0; i < ntokens; ++i)
        if (walks[i] >= 0) ++counts[walks[i]];
}

/* InitUnigramTable: count^0.75 normalised, Vose in fp64, same stack discipline. */
void oracle_unigram_build(int64_t n, const int32_t *counts, double *U, int32_t *K)
{
    double total = 0.0;
    for (int64_t i = 0; i < n; ++i) { U[i] = pow((double)counts[i], 0.75); total += U[i]; K[i] = 0; }
    for (int64_t i = 0; i < n; ++i) U[i] /= total;
    int32_t *small = (int32_t *)malloc(sizeof(int32_t) * (size_t)n), *large = (int32_t *)malloc(sizeof(int32_t) * (size_t)n);
    int64_t ns = 0, nl = 0;
    for (int64_t i = 0; i < n; ++i) {
        U[i] = U[i] * (double)n;
        if (U[i] < 1.0) small[ns++] = (int32_t)i; else large[nl++] = (int32_t)i;
    }
    while (ns > 0 && nl > 0) {
        const int32_t s = small[--ns], l = large[--nl];
        K[s] = l;
        U[l] = U[l] + U[s] - 1.0;
        if (U[l] < 1.0) small[ns++] = l; else large[nl++] = l;
    }
    while (ns > 0) U[small[--ns]] = 1.0;
    while (nl > 0) U[large[--nl]] = 1.0;
    free(small); free(large);
}

/* ---------------------------------------------------------------- SGNS (fp32)
 * TrainModel over walks [0, nwalks) in order, single thread (the sequential meaning of
 * the reference loop).  SynPos/SynNeg: [n][d] fp32, updated in place.
 * flags bit0: tokens < 0 are skipped (only when pad=-1); bit1: SNAP's RndUnigramInt quirk.
 * alpha(t) = alpha0*(1 - t/(epochs*tokens_total+1)), floor alpha0*1e-4, refreshed when
 * t % 10000 == 0 where t = token_offset + running count  (TrainModel's WordCntAll). */
static float sgns_alpha(float alpha0, int64_t t, int64_t denom)
{
    const int64_t tq = t - (t % 10000);
    float a = alpha0 * (1.0f - (float)((double)tq / (double)denom));
    if (a < alpha0 * 0.0001f) a = alpha0 * 0.0001f;
    return a;
}

/* The two lookups of RndUnigramInt are kept apart -- slot_tab[floor(u * n_slots)] names the entry X, (UT[X], KT[X]) decide the target -- because the
 * binary's own table layout (LearnVocab @0x40d560 renames the tokens by first appearance; gemhip_n2v_build_unigram_vocab_order, flag 16) stores the
 * first in slot space and the second in node space.  Node-id layout: slot_tab == KT, n_slots == n. */
static void sgns_train_core(int64_t n_slots, const int32_t *slot_tab, int32_t d, int64_t nwalks, int32_t walk_len, const int32_t *walks, int32_t window,
                            int32_t neg, float alpha0, int32_t epochs, int32_t epoch, int64_t tokens_total,
                            int64_t token_offset, int64_t walk_id_offset, const float *UT, const int32_t *KT, uint64_t seed,
                            int32_t flags, float *SynPos, float *SynNeg)
{
    (void)flags;
    float *neu1e = (float *)malloc(sizeof(float) * (size_t)d);
    const int64_t denom = (int64_t)epochs * tokens_total + 1;
    for (int64_t wl = 0; wl < nwalks; ++wl) {
        const int32_t *walk = walks + wl * walk_len;
        const int64_t wid = walk_id_offset + wl;
        for (int32_t pos = 0; pos < walk_len; ++pos) {
            const int64_t t = token_offset + wl * walk_len + pos;
            const float alpha = sgns_alpha(alpha0, t, denom);
            const int32_t word = walk[pos];
            if (word < 0) continue;
            const u32x4 rw = philox(seed, (uint32_t)wid, (uint32_t)((uint64_t)wid >> 32), (uint32_t)pos, TAG_WIN | ((uint32_t)epoch << 8));
            const int32_t b = (int32_t)(rw.x % (uint32_t)window);
            for (int32_t a = b; a < window * 2 + 1 - b; ++a) {
                if (a == window) continue;
                const int32_t cp = pos - window + a;
                if (cp < 0 || cp >= walk_len) continue;
                const int32_t ctx = walk[cp];
                if (ctx < 0) continue;
                float *xc = SynPos + (size_t)ctx * d;
                for (int32_t k = 0; k < d; ++k) neu1e[k] = 0.0f;
                for (int32_t j = 0; j < neg + 1; ++j) {
                    int32_t target; float label;
                    if (j == 0) { target = word; label = 1.0f; }
                    else {
                        const u32x4 rn = philox(seed, (uint32_t)wid, (uint32_t)((uint64_t)wid >> 32),
                                                (uint32_t)pos | ((uint32_t)a << 16), TAG_NEG | ((uint32_t)epoch << 8) | ((uint32_t)j << 16));
                        const uint32_t slot = mulhi_range(rn.x, (uint32_t)n_slots);
                        const int32_t X = slot_tab ? slot_tab[slot] : (int32_t)slot;
                        target = (u01(rn.y) < UT[X]) ? X : KT[X];
                        if (target == word) continue;
                        label = 0.0f;
                    }
                    float *yt = SynNeg + (size_t)target * d;
                    float f = 0.0f;
                    for (int32_t k = 0; k < d; ++k) f += xc[k] * yt[k];
                    float g;
                    if (f > 6.0f) g = (label - 1.0f) * alpha;
                    else if (f < -6.0f) g = label * alpha;
                    else g = (label - 1.0f + 1.0f / (1.0f + expf(f))) * alpha;
                    for (int32_t k = 0; k < d; ++k) {
                        neu1e[k] += g * yt[k];
                        yt[k] += g * xc[k];
                    }
                }
                for (int32_t k = 0; k < d; ++k) xc[k] += neu1e[k];
            }
        }
    }
    free(neu1e);
}

void oracle_sgns_train(int64_t n, int32_t d, int64_t nwalks, int32_t walk_len, const int32_t *walks, int32_t window,
                       int32_t neg, float alpha0, int32_t epochs, int32_t epoch, int64_t tokens_total,
                       int64_t token_offset, int64_t walk_id_offset, const float *UT, const int32_t *KT, uint64_t seed,
                       int32_t flags, float *SynPos, float *SynNeg)
{
    sgns_train_core(n, (flags & 2) ? KT : NULL, d, nwalks, walk_len, walks, window, neg, alpha0, epochs, epoch, tokens_total, token_offset, walk_id_offset,
                    UT, KT, seed, flags, SynPos, SynNeg);
}

/* TrainModel with the unigram table in the binary's layout: n_slots = nodes that occur, slot_tab[slot] = the node the slot names (the node of
 * KTable'[slot] under the RndUnigramInt quirk, else the slot's own node), UT / KT indexed by node (KT: the alias as a NODE). */
void oracle_sgns_train_vocab_order(int64_t n_slots, const int32_t *slot_tab, int32_t d, int64_t nwalks, int32_t walk_len, const int32_t *walks,
                                   int32_t window, int32_t neg, float alpha0, int32_t epochs, int32_t epoch, int64_t tokens_total,
                                   int64_t token_offset, int64_t walk_id_offset, const float *UT, const int32_t *KT, uint64_t seed,
                                   int32_t flags, float *SynPos, float *SynNeg)
{
    sgns_train_core(n_slots, slot_tab, d, nwalks, walk_len, walks, window, neg, alpha0, epochs, epoch, tokens_total, token_offset, walk_id_offset,
                    UT, KT, seed, flags, SynPos, SynNeg);
}

/* The SAME TrainModel restatement with the dot product X_ctx . Y_target accumulated in 32 interleaved partial sums (k mod 32) instead of one
 * running sum -- the only arithmetic difference from sgns_train_core (the device sums it across 64 lanes, a reordering of the same class); the row
 * updates, the sigmoid, every draw and the order of the pairs are those of sgns_train_core.  It exists because the strict loop's 128 dependent
 * additions per target make an R-MAT scale 22 pass (3.2e9 tokens) a 15-hour run; this one also computes a pair's six targets before it touches them
 * and prefetches their rows.  tests/test_oracle_n2v.py ties it to the strict function (max |difference| after a small pass at fp32 rounding level);
 * goldens made with it say so in their `engine`.  d must be a multiple of 32.  slot_tab as in sgns_train_core (NULL: slot == entry). */
typedef float v8f __attribute__((vector_size(32), aligned(4)));
/* the six targets of the pair (centre position pos, window slot a) -- tg[0] the centre word, tg[1..neg] the negatives, -1 where TrainModel skips
 * (Target == Word) -- with their SynNeg rows prefetched */
static inline void wide_targets(int64_t n_slots, const int32_t *slot_tab, const float *UT, const int32_t *KT, uint64_t seed, int64_t wid, int32_t pos,
                                int32_t a, int32_t epoch, int32_t neg, int32_t word, int32_t d, const float *SynNeg, int32_t *tg)
{
    tg[0] = word;
    for (int32_t j = 1; j < neg + 1; ++j) {
        const u32x4 rn = philox(seed, (uint32_t)wid, (uint32_t)((uint64_t)wid >> 32),
                                (uint32_t)pos | ((uint32_t)a << 16), TAG_NEG | ((uint32_t)epoch << 8) | ((uint32_t)j << 16));
        const uint32_t slot = mulhi_range(rn.x, (uint32_t)n_slots);
        const int32_t X = slot_tab ? slot_tab[slot] : (int32_t)slot;
        const int32_t target = (u01(rn.y) < UT[X]) ? X : KT[X];
        tg[j] = (target == word) ? -1 : target;
        if (tg[j] >= 0) {
            const char *q = (const char *)(SynNeg + (size_t)target * d);
            for (int32_t c = 0; c < d * 4; c += 64) __builtin_prefetch(q + c, 1, 1);
        }
    }
}

__attribute__((target("avx2")))
int32_t oracle_sgns_train_wide(int64_t n_slots, const int32_t *slot_tab, int32_t d, int64_t nwalks, int32_t walk_len, const int32_t *walks,
                               int32_t window, int32_t neg, float alpha0, int32_t epochs, int32_t epoch, int64_t tokens_total,
                               int64_t token_offset, int64_t walk_id_offset, const float *UT, const int32_t *KT, uint64_t seed,
                               int32_t flags, float *SynPos, float *SynNeg)
{
    (void)flags;
    if (d % 32 != 0 || neg > 15 || window > 4096) return -1;
    float *neu1e = (float *)aligned_alloc(64, sizeof(float) * (size_t)d);
    int32_t *slots = (int32_t *)malloc(sizeof(int32_t) * (size_t)(2 * window + 2));
    const int64_t denom = (int64_t)epochs * tokens_total + 1;
    for (int64_t wl = 0; wl < nwalks; ++wl) {
        const int32_t *walk = walks + wl * walk_len;
        const int64_t wid = walk_id_offset + wl;
        if (wl + 1 < nwalks) __builtin_prefetch(walk + walk_len, 0, 1);
        for (int32_t pos = 0; pos < walk_len; ++pos) {
            const int64_t t = token_offset + wl * walk_len + pos;
            const float alpha = sgns_alpha(alpha0, t, denom);
            const int32_t word = walk[pos];
            if (word < 0) continue;
            const u32x4 rw = philox(seed, (uint32_t)wid, (uint32_t)((uint64_t)wid >> 32), (uint32_t)pos, TAG_WIN | ((uint32_t)epoch << 8));
            const int32_t b = (int32_t)(rw.x % (uint32_t)window);
            /* the window slots this centre trains on, in TrainModel's order; the pair AFTER the one being trained has its targets drawn and its rows
             * prefetched already (the draws are counter-based: computing them early changes nothing) */
            int32_t np = 0;
            for (int32_t a = b; a < window * 2 + 1 - b; ++a) {
                if (a == window) continue;
                const int32_t cp = pos - window + a;
                if (cp < 0 || cp >= walk_len || walk[cp] < 0) continue;
                slots[np++] = a;
            }
            int32_t tga[16], tgb[16];
            int32_t *tg = tga, *tgn = tgb;
            if (np > 0) wide_targets(n_slots, slot_tab, UT, KT, seed, wid, pos, slots[0], epoch, neg, word, d, SynNeg, tg);
            for (int32_t ip = 0; ip < np; ++ip) {
                const int32_t a = slots[ip];
                const int32_t ctx = walk[pos - window + a];
                float *xc = SynPos + (size_t)ctx * d;
                if (ip + 1 < np) {
                    wide_targets(n_slots, slot_tab, UT, KT, seed, wid, pos, slots[ip + 1], epoch, neg, word, d, SynNeg, tgn);
                    const char *q = (const char *)(SynPos + (size_t)walk[pos - window + slots[ip + 1]] * d);
                    for (int32_t c = 0; c < d * 4; c += 64) __builtin_prefetch(q + c, 1, 1);
                }
                for (int32_t k = 0; k < d; k += 8) *(v8f *)(neu1e + k) = (v8f){0, 0, 0, 0, 0, 0, 0, 0};
                for (int32_t j = 0; j < neg + 1; ++j) {
                    if (tg[j] < 0) continue;
                    const float label = j == 0 ? 1.0f : 0.0f;
                    float *yt = SynNeg + (size_t)tg[j] * d;
                    v8f a0 = {0, 0, 0, 0, 0, 0, 0, 0}, a1 = a0, a2 = a0, a3 = a0;
                    for (int32_t k = 0; k < d; k += 32) {
                        a0 += *(const v8f *)(xc + k) * *(const v8f *)(yt + k);
                        a1 += *(const v8f *)(xc + k + 8) * *(const v8f *)(yt + k + 8);
                        a2 += *(const v8f *)(xc + k + 16) * *(const v8f *)(yt + k + 16);
                        a3 += *(const v8f *)(xc + k + 24) * *(const v8f *)(yt + k + 24);
                    }
                    const v8f s = (a0 + a1) + (a2 + a3);
                    const float f = ((s[0] + s[1]) + (s[2] + s[3])) + ((s[4] + s[5]) + (s[6] + s[7]));
                    float g;
                    if (f > 6.0f) g = (label - 1.0f) * alpha;
                    else if (f < -6.0f) g = label * alpha;
                    else g = (label - 1.0f + 1.0f / (1.0f + expf(f))) * alpha;
                    const v8f gv = {g, g, g, g, g, g, g, g};
                    for (int32_t k = 0; k < d; k += 8) {
                        const v8f y = *(const v8f *)(yt + k);
                        *(v8f *)(neu1e + k) += gv * y;
                        *(v8f *)(yt + k) = y + gv * *(const v8f *)(xc + k);
                    }
                }
                for (int32_t k = 0; k < d; k += 8) *(v8f *)(xc + k) += *(const v8f *)(neu1e + k);
                int32_t *sw = tg; tg = tgn; tgn = sw;
            }
        }
    }
    free(neu1e); free(slots);
    return 0;
}

void oracle_sgns_init(int64_t n, int32_t d, uint64_t seed, float *SynPos, float *SynNeg)
{
    const int64_t total = n * (int64_t)d;
    for (int64_t t = 0; t * 4 < total; ++t) {
        const u32x4 r = philox(seed, (uint32_t)t, (uint32_t)((uint64_t)t >> 32), 0u, TAG_INIT);
        const uint32_t v[4] = {r.x, r.y, r.z, r.w};
        for (int k = 0; k < 4; ++k)
            if (t * 4 + k < total) SynPos[t * 4 + k] = (u01(v[k]) - 0.5f) / (float)d;
    }
    memset(SynNeg, 0, sizeof(float) * (size_t)total);
}

/* ------------------------------------------------------------------------------------------------
 * Partitioned ("episode") SGNS -- restatement of the multi-GPU schedule of gem_amd/multi_gpu.py, NOT of a
 * reference function (the reference is single-process).  Nodes are split into `parts` partitions
 * (node v -> v % parts); the (context, word) pairs TrainModel forms fall into parts x parts buckets by
 * (part(context), part(word)); in round s partition g of SynPos meets partition (g+s)%parts of SynNeg, so no
 * two workers ever touch the same row.  oracle_sgns_pairs lists the pairs (tests count them);
 * oracle_sgns_train_part below trains one bucket in walk order. */
int64_t oracle_sgns_pairs(int64_t nwalks, int32_t walk_len, const int32_t *walks, int32_t window, int32_t epoch,
                          int64_t walk_id_offset, uint64_t seed, int32_t *ctx_out, int32_t *word_out /* may be NULL: count only */)
{
    int64_t np = 0;
    for (int64_t wl = 0; wl < nwalks; ++wl) {
        const int32_t *walk = walks + wl * walk_len;
        const int64_t wid = walk_id_offset + wl;
        for (int32_t pos = 0; pos < walk_len; ++pos) {
            const int32_t word = walk[pos];
            if (word < 0) continue;
            const u32x4 rw = philox(seed, (uint32_t)wid, (uint32_t)((uint64_t)wid >> 32), (uint32_t)pos, TAG_WIN | ((uint32_t)epoch << 8));
            const int32_t b = (int32_t)(rw.x % (uint32_t)window);
            for (int32_t a = b; a < window * 2 + 1 - b; ++a) {
                if (a == window) continue;
                const int32_t cp = pos - window + a;
                if (cp < 0 || cp >= walk_len) continue;
                if (walk[cp] < 0) continue;
                if (ctx_out) { ctx_out[np] = walk[cp]; word_out[np] = word; }
                ++np;
            }
        }
    }
    return np;
}

/* One bucket of the partitioned schedule in WALK order (round 4, gemhip_sgns_train_part / sgns_win_kernel<PART>): oracle_sgns_train above --
 * TrainModel, ELF @0x40d6a0 -- restricted to the pairs whose context is a node of partition ctx_part and whose centre word is a node of partition
 * word_part (node v -> partition v % parts); negatives from the unigram table restricted to word_part (UTp / KTp over local indices, global id =
 * local * parts + word_part).  Same Philox keys per (walk id, position, slot) as the unrestricted function, so parts = 1 reproduces it exactly.
 * wids: global walk id of every walk in the buffer (NULL: walk_id_offset + index).  local_rows == 0: SynPos / SynNeg are the FULL tables (global
 * rows); != 0: they are the partition buffers of ctx_part / word_part (row v / parts).  Returns the number of (centre, context) pairs trained. */
/* ... with the partition's table in the binary's layout (oracle_sgns_train_part_slots): n_slots / slot_tab as in sgns_train_core -- the slot
 * floor(u * n_slots) names the LOCAL entry slot_tab[slot], (UTp[X], KTp[X]) over local indices decide the target.  slot_tab == NULL: the node-id layout
 * (n_slots = n_local, X = KTp[slot] under flags & 2, else the slot itself). */
int64_t oracle_sgns_train_part_slots(int32_t d, int64_t nwalks, int32_t walk_len, const int32_t *walks, const int64_t *wids, int64_t walk_id_offset,
                            int32_t window, float alpha0, int64_t alpha_tokens_total, int64_t token_offset, int32_t epoch, int32_t parts,
                            int32_t ctx_part, int32_t word_part, int64_t n_local, const float *UTp, const int32_t *KTp, int64_t n_slots,
                            const int32_t *slot_tab, uint64_t seed, int32_t flags, int32_t local_rows, float *SynPos, float *SynNeg);

int64_t oracle_sgns_train_part(int32_t d, int64_t nwalks, int32_t walk_len, const int32_t *walks, const int64_t *wids, int64_t walk_id_offset,
                            int32_t window, float alpha0, int64_t alpha_tokens_total, int64_t token_offset, int32_t epoch, int32_t parts,
                            int32_t ctx_part, int32_t word_part, int64_t n_local, const float *UTp, const int32_t *KTp, uint64_t seed,
                            int32_t flags, int32_t local_rows, float *SynPos, float *SynNeg)
{
    return oracle_sgns_train_part_slots(d, nwalks, walk_len, walks, wids, walk_id_offset, window, alpha0, alpha_tokens_total, token_offset, epoch, parts,
                                        ctx_part, word_part, n_local, UTp, KTp, n_local, NULL, seed, flags, local_rows, SynPos, SynNeg);
}

int64_t oracle_sgns_train_part_slots(int32_t d, int64_t nwalks, int32_t walk_len, const int32_t *walks, const int64_t *wids, int64_t walk_id_offset,
                            int32_t window, float alpha0, int64_t alpha_tokens_total, int64_t token_offset, int32_t epoch, int32_t parts,
                            int32_t ctx_part, int32_t word_part, int64_t n_local, const float *UTp, const int32_t *KTp, int64_t n_slots,
                            const int32_t *slot_tab, uint64_t seed, int32_t flags, int32_t local_rows, float *SynPos, float *SynNeg)
{
    (void)n_local;
    float *neu1e = (float *)malloc(sizeof(float) * (size_t)d);
    const int64_t denom = alpha_tokens_total + 1;
    int64_t npairs = 0;
    for (int64_t wl = 0; wl < nwalks; ++wl) {
        const int32_t *walk = walks + wl * walk_len;
        const int64_t wid = wids ? wids[wl] : walk_id_offset + wl;
        for (int32_t pos = 0; pos < walk_len; ++pos) {
            const int64_t t = token_offset + wl * walk_len + pos;
            const float alpha = sgns_alpha(alpha0, t, denom);
            const int32_t word = walk[pos];
            if (word < 0 || word % parts != word_part) continue;
            const u32x4 rw = philox(seed, (uint32_t)wid, (uint32_t)((uint64_t)wid >> 32), (uint32_t)pos, TAG_WIN | ((uint32_t)epoch << 8));
            const int32_t b = (int32_t)(rw.x % (uint32_t)window);
            for (int32_t a = b; a < window * 2 + 1 - b; ++a) {
                if (a == window) continue;
                const int32_t cp = pos - window + a;
                if (cp < 0 || cp >= walk_len) continue;
                const int32_t ctx = walk[cp];
                if (ctx < 0 || ctx % parts != ctx_part) continue;
                ++npairs;
                float *xc = SynPos + (size_t)(local_rows ? ctx / parts : ctx) * d;
                for (int32_t k = 0; k < d; ++k) neu1e[k] = 0.0f;
                for (int32_t j = 0; j < 6; ++j) {
                    int32_t target; float label;
                    if (j == 0) { target = word; label = 1.0f; }
                    else {
                        const u32x4 rn = philox(seed, (uint32_t)wid, (uint32_t)((uint64_t)wid >> 32),
                                                (uint32_t)pos | ((uint32_t)a << 16), TAG_NEG | ((uint32_t)epoch << 8) | ((uint32_t)j << 16));
                        const uint32_t slot = mulhi_range(rn.x, (uint32_t)n_slots);
                        const int32_t X = slot_tab ? slot_tab[slot] : ((flags & 2) ? KTp[slot] : (int32_t)slot);
                        const int32_t loc = (u01(rn.y) < UTp[X]) ? X : KTp[X];
                        target = loc * parts + word_part;
                        if (target == word) continue;
                        label = 0.0f;
                    }
                    float *yt = SynNeg + (size_t)(local_rows ? target / parts : target) * d;
                    float f = 0.0f;
                    for (int32_t k = 0; k < d; ++k) f += xc[k] * yt[k];
                    float g;
                    if (f > 6.0f) g = (label - 1.0f) * alpha;
                    else if (f < -6.0f) g = label * alpha;
                    else g = (label - 1.0f + 1.0f / (1.0f + expf(f))) * alpha;
                    for (int32_t k = 0; k < d; ++k) { neu1e[k] += g * yt[k]; yt[k] += g * xc[k]; }
                }
                for (int32_t k = 0; k < d; ++k) xc[k] += neu1e[k];
            }
        }
    }
    free(neu1e);
    return npairs;
}
