// sgns.hpp -- skip-gram with negative sampling on gfx950: TrainModel of the SNAP node2vec binary (ELF @0x40d6a0) as device code.
// Shared by the three translation units that carry it:
//   sgns_hogwild.hip  sgns_win_kernel<.., DELTA = true, ..>   the shipped Hogwild path (delta write-back, reload-on-update, hot rows)
//   sgns_det.hip      sgns_win_kernel<.., DELTA = false, ..> + sgns_kernel   deterministic / single-wavefront launches and d >= 384
//   sgns_part.hip     sgns_win_kernel<.., PART = true>   one bucket (SynPos partition g x SynNeg partition h) of the partitioned N-GPU schedule
//   n2v.hip           the C ABI (gemhip_sgns_train / gemhip_sgns_train_part pick a launcher), walks, alias tables
// Every kernel template is instantiated in exactly one of them; the helpers below are inlined wherever they are used.
#pragma once
#include "common.hpp"
#include <type_traits>
#include <cstdint>

namespace gemhip {

struct SgnsArgs {
    const int32_t *walks; int64_t walk_lo, walk_hi; int32_t walk_len; int32_t window;
    float alpha0; int64_t denom; int64_t token_offset; int64_t walk_id_offset; int32_t epoch;
    // negative sampling (RndUnigramInt): slot = floor(u n) -> X = KT[slot] (flags & 2; else X = slot) -> target = u' < UK[X].x ? X : UK[X].y.  KT is indexed
    // by SLOT, UK = {UTable, KTable} by NODE: the same arrays in the node-id layout, different ones in the binary's vocabulary-order layout (n2v.hip)
    const float *UT; const int32_t *KT; const uint2 *UK; uint32_t n; uint64_t seed; int32_t flags; int32_t d;
    // sgns_win_kernel: the same draw through ONE gather.  SK[slot] = {X, bits of UTable[X], KTable[X], 0} with X = KT[slot] (flags & 2) or slot: what the
    // two dependent gathers KT[slot] -> UK[X] return, laid out by slot (n2v.hip: n2v_slot_table_kernel, built once per table and quirk setting).  One
    // 16-byte random read per draw instead of a 4-byte and an 8-byte one: half the table sectors per pair (5 x 64 B less of ~6.5 KB, the counters
    // charge every sector) and one dependent round trip less in the negative-target pipeline
    const uint4 *SK;
    float *SynPos; float *SynNeg; int32_t nwaves; unsigned long long *pairs;
    float *dummy;               // sgns_win_kernel: nwaves rows, never read for their value
    unsigned long long *prof;   // GEMHIP_SGNS_PROFILE builds only: per-phase cycle sums (s_memtime)
    int32_t cache_radius;       // sgns_win_kernel: tokens within this many positions of the centre keep their SynPos row in LDS
    int32_t prefetch;           // sgns_win_kernel: pairs whose negative rows are requested ahead (2, or 1)
    int32_t reload;             // sgns_win_kernel<RELOAD>: negative rows updated as they are at store time, centre row by atomic add
    const int32_t *counts; int32_t hot_thr;   // sgns_win_kernel<!ALLC>: nodes with counts[v] >= hot_thr > 0 never enter the LDS window (HOT ROWS below); `counts` is the
                                              // token count, or -- single-GPU Hogwild launches -- n2v.hip's hotkey: INT32_MAX for a LOCALLY hot node (tokens packed into few walks)
    // sgns_win_kernel<PART> (partitioned tables, N-GPU schedule): node v belongs to partition v % parts, local row v / parts.  SynPos / SynNeg point
    // at partition ctx_part of SynPos and partition word_part of SynNeg; UT / KT / UK / n describe the unigram table RESTRICTED to word_part (local
    // indices); `counts` stays global.  Only pairs (context in ctx_part, centre word in word_part) are trained: TrainModel filtered to one bucket.
    int32_t parts, ctx_part, word_part;
    // a corpus assembled from several ranks' walk shards: work item wl = r * seg_len + j is walk j of segment r, stored at row seg[r] + j of
    // `walks`, present when j < seg[nseg + r], with global walk id (the Philox key) seg[2 * nseg + r] + j.  seg == nullptr: one segment, row = wl,
    // walk id = walk_id_offset + wl
    const int64_t *seg; int32_t nseg; int64_t seg_len;
    float *scratch;             // sgns_win_kernel<PART, DELTA>: nwaves x (2R+1) rows -- the window rows as loaded (delta write-back), kept out of LDS
    // FRESH HOT ROWS (round 6; Hogwild launches that have hot rows).  bit 0: a hot centre word's positive row SynNeg[word] takes every pair's update as a
    // RETURNING atomic add -- the next pair computes with the row as memory held it one pair step ago (what came back + this pair's own change), not with
    // the copy loaded at the centre's start, ~10 pair steps old by the centre's end.  bit 1: hot negative rows are fetched AGAIN right before the dot
    // products, so the gradient is computed from the row as it is now and not from the copy requested two pairs ahead.
    // bit 2 (value 4): EVERY negative row with at least neg_thr tokens takes its update as an atomic add (as the hot rows do), not as reload + store.  The
    // unigram^0.75 draw -- under RndUnigramInt's quirk: over the alias targets only -- concentrates on a few thousand mid-frequency rows that are "cold" by
    // the window's measure (token count) but are touched by another wavefront inside one's load..store interval a third of the time at 768 wavefronts
    // (profiles/r06_staleness_rmat17.jsonl): the reload + store then overwrites that update.
    int32_t fresh; int32_t neg_thr;
    unsigned int *stale_ver; unsigned long long *stale_hist; int64_t n_nodes;   // GEMHIP_SGNS_STALENESS builds only (see STALE below)
};

using sgns_fn = void (*)(const SgnsArgs &, int blocks, int threads, size_t lds, hipStream_t);
sgns_fn pick_sgns(int d);                    // sgns_det.hip: sgns_kernel (no LDS window), nullptr when d is unsupported (even d <= 512, odd d <= 256)
sgns_fn pick_sgns_win_det(int d);            // sgns_det.hip: sgns_win_kernel, overwrite on leave (bit-compatible with sgns_kernel on one wavefront)
sgns_fn pick_sgns_win_hogwild(int d);        // sgns_hogwild.hip: sgns_win_kernel, delta write-back
sgns_fn pick_sgns_win_part(int d, bool hogwild);   // sgns_part.hip: sgns_win_kernel<PART> (one bucket of the partitioned schedule), Hogwild or single-wavefront
// floats one cached row occupies in LDS and in the per-wave scratch row: RW = NV * VEC * 64 of the kernel the pickers INSTANTIATE for d
// (NV is 1, 2 or 4: three chunks run on the NV = 4 kernel, so d = 129..191 odd and 258..384 even occupy 256 / 512 floats, not 192 / 384;
// round 3 sized LDS and the scratch rows from the chunk count and the NV = 4 kernels wrote past both -- ADVICE r3)
inline int sgns_win_nv(int d) { const int nv = d % 2 == 0 ? (d + 127) / 128 : (d + 63) / 64; return nv <= 1 ? 1 : nv <= 2 ? 2 : 4; }
inline int sgns_win_row_floats(int d) { return sgns_win_nv(d) * (d % 2 == 0 ? 128 : 64); }
}  // namespace gemhip

using namespace gemhip;

namespace {
enum { TAG_WALK = 1, TAG_WIN = 2, TAG_NEG = 3, TAG_INIT = 4 };
constexpr int SGNS_NEG = 5;             // SNAP: NegSamN = 5 (compile-time constant there too)
constexpr float SGNS_MAX_EXP = 6.0f;    // SNAP: MaxExp
constexpr int HOGWILD_ROWS_PER_WAVE = 128;

__host__ __device__ __forceinline__ uint32_t mulhi_range(uint32_t r, uint32_t n) { return (uint32_t)(((uint64_t)r * n) >> 32); }

__host__ __device__ __forceinline__ uint32_t fmix32(uint32_t h)
{
    h ^= h >> 16; h *= 0x85EBCA6Bu; h ^= h >> 13; h *= 0xC2B2AE35u; h ^= h >> 16;
    return h;
}


// Embedding rows are shared, concurrently updated state (Hogwild).  gfx950 has one L1 per CU that is
// never refreshed by other CUs' stores and one write-back L2 per XCD that is not coherent with the
// other seven, so PLAIN loads/stores let every CU train on its own stale copy of the table (measured:
// SBM-1024 MAP 0.177 on one CU -> 0.09 on many).  All row traffic therefore uses relaxed AGENT-scope
// atomic accesses (global_load/store ... sc1): they bypass L1, are coherent across XCDs per location,
// and cost the same bytes.  8 bytes per lane when d is even, 4 otherwise.
using gu64 = __attribute__((address_space(1))) unsigned long long;
using gu32 = __attribute__((address_space(1))) unsigned int;

template <int VEC>
__device__ __forceinline__ void ld_row(const float *p, int d, int lane, int c, float (&v)[VEC])
{
    const int idx = (c * WAVE + lane) * VEC;
    if constexpr (VEC == 2) {
        if (idx < d) {
            const unsigned long long t = __hip_atomic_load((gu64 *)(p + idx), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            v[0] = __builtin_bit_cast(float, (unsigned int)t);
            v[1] = __builtin_bit_cast(float, (unsigned int)(t >> 32));
        } else { v[0] = 0.f; v[1] = 0.f; }
    } else {
        v[0] = idx < d ? __builtin_bit_cast(float, __hip_atomic_load((gu32 *)(p + idx), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) : 0.f;
    }
}
template <int VEC>
__device__ __forceinline__ void st_row(float *p, int d, int lane, int c, const float (&v)[VEC])
{
    const int idx = (c * WAVE + lane) * VEC;
    if (idx < d) {
        if constexpr (VEC == 2) {
            const unsigned long long t = (unsigned long long)__builtin_bit_cast(unsigned int, v[0]) |
                                         ((unsigned long long)__builtin_bit_cast(unsigned int, v[1]) << 32);
            __hip_atomic_store((gu64 *)(p + idx), t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else {
            __hip_atomic_store((gu32 *)(p + idx), __builtin_bit_cast(unsigned int, v[0]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

// this lane's VEC floats of a row, to an address the caller already holds (same relaxed agent-scope store as st_row)
template <int VEC>
__device__ __forceinline__ void st_lane(float *q, const float (&v)[VEC])
{
    if constexpr (VEC == 2) {
        const unsigned long long t = (unsigned long long)__builtin_bit_cast(unsigned int, v[0]) | ((unsigned long long)__builtin_bit_cast(unsigned int, v[1]) << 32);
        __hip_atomic_store((gu64 *)q, t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else __hip_atomic_store((gu32 *)q, __builtin_bit_cast(unsigned int, v[0]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}


// gradient scale of TrainModel: (label - sigma(f)) * alpha with the +-MaxExp clamps
__device__ __forceinline__ float sgns_grad(float f, float label, float alpha)
{
    if (f > SGNS_MAX_EXP) return (label - 1.0f) * alpha;
    if (f < -SGNS_MAX_EXP) return label * alpha;
    return (label - 1.0f + 1.0f / (1.0f + expf(f))) * alpha;
}

// The same quantity without branches and with the hardware's 1-ulp exp / reciprocal (v_exp_f32, v_rcp_f32) instead of the IEEE expf and
// division sequences (~45 instructions, three exec-mask branches): (label - sigma(f)) * alpha, sigma forced to 1 / 0 beyond +-MaxExp exactly
// like TrainModel's clamps.  Differs from sgns_grad by ~2e-7 relative -- three orders below the 2e-4 bar against the oracle (whose own
// reference, SNAP, reads sigma from a 1000-entry table).  Used by the window kernels' lane-parallel sigmoid.
__device__ __forceinline__ float sgns_grad_fast(float f, float label, float alpha)
{
    const float e = __expf(f);                                    // v_exp_f32(f * log2 e)
    float one_minus_sigma = __builtin_amdgcn_rcpf(1.0f + e);      // 1 / (1 + e^f) = 1 - sigma(f)  (v_rcp_f32, 1 ulp; __frcp_rn compiles to the 10-instruction IEEE division)
    one_minus_sigma = f > SGNS_MAX_EXP ? 0.0f : one_minus_sigma;
    one_minus_sigma = f < -SGNS_MAX_EXP ? 1.0f : one_minus_sigma;
    return (label - 1.0f + one_minus_sigma) * alpha;
}

// TrainModel (ELF @0x40d6a0).  One wavefront owns one walk: tokens and the pre-drawn
// negative targets of the current centre sit in LDS; the centre's positive row SynNeg[word]
// stays in registers across all its contexts; per context the context row and the five
// negative rows are fetched together (6 coalesced 4d-byte reads in flight), reduced with
// DPP wave sums, and written back.  Hogwild across wavefronts, exactly sequential inside one.
template <int VEC, int NV>
__global__ __launch_bounds__(256) void sgns_kernel(SgnsArgs A)
{
    extern __shared__ __attribute__((aligned(16))) int32_t lds[];
    const int lane = lane_id();
    const int wave = threadIdx.x >> 6;
    const int per_wave = A.walk_len + 2 * A.window * SGNS_NEG;
    int32_t *tok = lds + wave * per_wave;
    int32_t *negs = tok + A.walk_len;
    const int64_t gw = (int64_t)blockIdx.x * (blockDim.x >> 6) + wave;
    if (gw >= A.nwaves) return;
    const int d = A.d;
    const int win = A.window;
    const bool quirk = (A.flags & 2) != 0;

    unsigned long long npairs = 0;
    for (int64_t wl = A.walk_lo + gw; wl < A.walk_hi; wl += A.nwaves) {
        const int32_t *walk = A.walks + wl * A.walk_len;
        for (int k = lane; k < A.walk_len; k += WAVE) tok[k] = walk[k];
        __builtin_amdgcn_wave_barrier();
        const int64_t wid = A.walk_id_offset + wl;
        const uint32_t w_lo = (uint32_t)wid, w_hi = (uint32_t)((uint64_t)wid >> 32);

        for (int pos = 0; pos < A.walk_len; ++pos) {
            const int32_t word = __builtin_amdgcn_readfirstlane(tok[pos]);
            if (word < 0) continue;
            // alpha: refreshed every 10000 words of the global count (TrainModel)
            const int64_t t = A.token_offset + wl * A.walk_len + pos;
            const int64_t tq = t - (t % 10000);
            float alpha = A.alpha0 * (1.0f - (float)((double)tq / (double)A.denom));
            alpha = fmaxf(alpha, A.alpha0 * 0.0001f);
            const u32x4 rw = philox4x32_10(A.seed, w_lo, w_hi, (uint32_t)pos, (uint32_t)TAG_WIN | ((uint32_t)A.epoch << 8));
            const int b = (int)(rw.x % (uint32_t)win);
            // draw every negative target of this centre at once: sample s = (a, j) -> lane-parallel table lookups
            const int nsamp = 2 * win * SGNS_NEG;
            for (int s = lane; s < nsamp; s += WAVE) {
                const int ai = s / SGNS_NEG;                 // 0 .. 2*win-1  (context slot, skipping the centre)
                const int a = ai < win ? ai : ai + 1;
                const int j = s - ai * SGNS_NEG + 1;
                const u32x4 rn = philox4x32_10(A.seed, w_lo, w_hi, (uint32_t)pos | ((uint32_t)a << 16),
                                               (uint32_t)TAG_NEG | ((uint32_t)A.epoch << 8) | ((uint32_t)j << 16));
                const uint32_t slot = mulhi_range(rn.x, A.n);
                const int32_t X = quirk ? A.KT[slot] : (int32_t)slot;          // RndUnigramInt (ELF @0x40d5f0); A.KT: the table indexed by SLOT
                const uint2 uk = A.UK[X];                                      // {UTable[X], KTable[X]} indexed by NODE (the two differ in the vocabulary-order layout)
                negs[s] = (u01(rn.y) < __builtin_bit_cast(float, uk.x)) ? X : (int32_t)uk.y;
            }
            __builtin_amdgcn_wave_barrier();

            float yp[NV][VEC];                               // SynNeg[word]: positive target of every context of this centre
            float *pp = A.SynNeg + (int64_t)word * d;
#pragma unroll
            for (int c = 0; c < NV; ++c) ld_row<VEC>(pp, d, lane, c, yp[c]);

            for (int a = b; a < 2 * win + 1 - b; ++a) {
                if (a == win) continue;
                const int cp = pos - win + a;
                if (cp < 0 || cp >= A.walk_len) continue;
                const int32_t ctx = __builtin_amdgcn_readfirstlane(tok[cp]);
                if (ctx < 0) continue;
                const int ai = a < win ? a : a - 1;
                ++npairs;
                int32_t tgt[SGNS_NEG];
#pragma unroll
                for (int j = 0; j < SGNS_NEG; ++j) tgt[j] = __builtin_amdgcn_readfirstlane(negs[ai * SGNS_NEG + j]);

                float xc[NV][VEC], neu[NV][VEC], yn[SGNS_NEG][NV][VEC];
                float *pc = A.SynPos + (int64_t)ctx * d;
#pragma unroll
                for (int c = 0; c < NV; ++c) ld_row<VEC>(pc, d, lane, c, xc[c]);
#pragma unroll
                for (int j = 0; j < SGNS_NEG; ++j) {
                    const float *pn = A.SynNeg + (int64_t)tgt[j] * d;
#pragma unroll
                    for (int c = 0; c < NV; ++c) ld_row<VEC>(pn, d, lane, c, yn[j][c]);
                }
#pragma unroll
                for (int c = 0; c < NV; ++c)
#pragma unroll
                    for (int v = 0; v < VEC; ++v) neu[c][v] = 0.f;

                {   // j = 0: positive target (label 1), row lives in registers
                    float part = 0.f;
#pragma unroll
                    for (int c = 0; c < NV; ++c)
#pragma unroll
                        for (int v = 0; v < VEC; ++v) part = fmaf(xc[c][v], yp[c][v], part);
                    const float g = sgns_grad(wave_sum(part), 1.0f, alpha);
#pragma unroll
                    for (int c = 0; c < NV; ++c)
#pragma unroll
                        for (int v = 0; v < VEC; ++v) { neu[c][v] = fmaf(g, yp[c][v], neu[c][v]); yp[c][v] = fmaf(g, xc[c][v], yp[c][v]); }
                }
#pragma unroll
                for (int j = 0; j < SGNS_NEG; ++j) {
                    if (tgt[j] == word) continue;                        // TrainModel: `if (Target == Word) continue`
                    // a target drawn twice for this context must see the first update (sequential semantics)
#pragma unroll
                    for (int jp = 0; jp < j; ++jp)
                        if (tgt[jp] == tgt[j] && tgt[jp] != word) {
#pragma unroll
                            for (int c = 0; c < NV; ++c)
#pragma unroll
                                for (int v = 0; v < VEC; ++v) yn[j][c][v] = yn[jp][c][v];
                        }
                    float part = 0.f;
#pragma unroll
                    for (int c = 0; c < NV; ++c)
#pragma unroll
                        for (int v = 0; v < VEC; ++v) part = fmaf(xc[c][v], yn[j][c][v], part);
                    const float g = sgns_grad(wave_sum(part), 0.0f, alpha);
#pragma unroll
                    for (int c = 0; c < NV; ++c)
#pragma unroll
                        for (int v = 0; v < VEC; ++v) { neu[c][v] = fmaf(g, yn[j][c][v], neu[c][v]); yn[j][c][v] = fmaf(g, xc[c][v], yn[j][c][v]); }
                    float *pn = A.SynNeg + (int64_t)tgt[j] * d;
#pragma unroll
                    for (int c = 0; c < NV; ++c) st_row<VEC>(pn, d, lane, c, yn[j][c]);
                }
#pragma unroll
                for (int c = 0; c < NV; ++c) {
#pragma unroll
                    for (int v = 0; v < VEC; ++v) xc[c][v] += neu[c][v];
                    st_row<VEC>(pc, d, lane, c, xc[c]);
                }
            }
#pragma unroll
            for (int c = 0; c < NV; ++c) st_row<VEC>(pp, d, lane, c, yp[c]);
            __builtin_amdgcn_wave_barrier();
        }
    }
    if (lane == 0 && A.pairs) atomicAdd(A.pairs, npairs);
}

#ifdef GEMHIP_SGNS_PROFILE
#define PROF_T() ({ asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); unsigned long long _t = __builtin_amdgcn_s_memtime(); asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); _t; })
#define PROF_DECL unsigned long long prof_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}; unsigned long long prof_t = 0
#define PROF_START() prof_t = PROF_T()
#define PROF_LAP(k) do { const unsigned long long _n = PROF_T(); prof_acc[k] += _n - prof_t; prof_t = _n; } while (0)
#define PROF_WAIT_VM(n) asm volatile("s_waitcnt vmcnt(" #n ")" ::: "memory")
#else
#define PROF_DECL
#define PROF_START()
#define PROF_LAP(k)
#define PROF_WAIT_VM(n)
#endif

// GEMHIP_SGNS_STALENESS builds (scripts/build_variant.sh stale -DGEMHIP_SGNS_STALENESS): HOW STALE is the copy a gradient is computed from?  Every row
// carries an update counter (stale_ver[t] for SynNeg[t], stale_ver[n + v] for SynPos[v], bumped by whoever applies an update); a wavefront notes the
// counter when it loads a row and again when it applies its update -- the difference is the number of FOREIGN updates the row took in between, i.e. the
// number of gradients that were computed concurrently from the same base.  Histogram stale_hist[class][floor(log2 count)][bit length of the difference]:
// class 0 a negative target, 1 the centre word's positive row (per pair), 2 an uncached (hot) context row.
#ifdef GEMHIP_SGNS_STALENESS
#define STALE(...) __VA_ARGS__
__device__ __forceinline__ void stale_rec(unsigned long long *hist, int cls, int cnt, unsigned int diff)
{
    const int hb = 31 - __clz(cnt > 1 ? cnt : 1), sb = diff == 0u ? 0 : (32 - __clz(diff) > 15 ? 15 : 32 - __clz(diff));
    atomicAdd(hist + (cls * 32 + hb) * 16 + sb, 1ull);
}
__device__ __forceinline__ unsigned int stale_load(const unsigned int *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
#else
#define STALE(...)
#endif

// ---- window-cached TrainModel (default since round 2) ------------------------------------------------------------
// Same arithmetic, same order and same Philox draws as sgns_kernel; what changes is WHERE the context rows live.
// A token is a context of every centre within `window` positions, so sgns_kernel moves its SynPos row 2 x ~11 times.
// Here one wavefront (= one 64-thread block = one walk at a time) keeps the rows of the tokens within `R` positions
// of the centre in LDS: a token's row is read once when it enters the window and written once when it leaves
// (2R+1 slots; repeated nodes share a slot through a (node -> slot) directory held across the lanes, so the sequence of
// values every row takes inside one wavefront is exactly TrainModel's).  Contexts farther than R (only possible when
// R < window) go straight to memory as before -- the directory lookup precedes every access, so cached and direct
// accesses never alias.
// Hogwild: other wavefronts may update a cached row while it sits in LDS.  DELTA mode (multi-wave launches) keeps the
// row as loaded next to the working copy and leaves with `row_now + (working - loaded)`: nothing another wavefront
// wrote in between is lost (the read-modify-write window is one centre step, as short as sgns_kernel's per-pair
// windows); a single-wave (deterministic) launch writes the working copy back as is.
// Latency at 1 wave per SIMD-ish occupancy (the LDS window bounds residency at ~7-13 waves per CU): the negative rows of
// the next TWO (centre, context) pairs are in flight while a pair is computed (targets equal to a row updated in
// between are re-forwarded from registers: exact), and the negative targets are drawn two centres ahead.
// Six dot products at once.  Every lane holds its partial sums p[0..5]; on return lane l holds the WAVE TOTAL of value
// idx(l) = (l & 4) ? 4 + (l & 1) : (l & 3)  (so lanes 0..5 hold totals 0..5).  Transposing while reducing (each lane keeps the half
// of the values its lane bit selects and hands the other half to its partner) costs 22 lane operations for all six sums,
// against 6 x 11 for six wave_sum calls -- and leaves the six totals in six LANES, so the sigmoid of TrainModel runs once,
// lane-parallel, instead of six times.
template <int CTRL>
__device__ __forceinline__ float dpp_mov(float v)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}
typedef unsigned int n2v_u32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float wave_sum6(const float (&p)[6], int lane)
{
    const bool b0 = (lane & 1) != 0, b1 = (lane & 2) != 0, b2 = (lane & 4) != 0;
    // lane bit 0 (partner l^1): values (0,1) (2,3) (4,5)
    const float a0 = (b0 ? p[1] : p[0]) + dpp_mov<0xB1>(b0 ? p[0] : p[1]);
    const float a1 = (b0 ? p[3] : p[2]) + dpp_mov<0xB1>(b0 ? p[2] : p[3]);
    const float a2 = (b0 ? p[5] : p[4]) + dpp_mov<0xB1>(b0 ? p[4] : p[5]);
    // lane bit 1 (partner l^2): quad totals; value index 2*b1 + b0 in c0, 4 + b0 in c1
    float c0 = (b1 ? a1 : a0) + dpp_mov<0x4E>(b1 ? a0 : a1);
    float c1 = a2 + dpp_mov<0x4E>(a2);
    // the four quads of a row of 16 lanes: rotate by 4 and by 8
    c0 += dpp_mov<0x124>(c0); c0 += dpp_mov<0x128>(c0);
    c1 += dpp_mov<0x124>(c1); c1 += dpp_mov<0x128>(c1);
    float m = b2 ? c1 : c0;
    // the four rows: v_permlane16_swap / v_permlane32_swap of (m, m) give {even-row copy, odd-row copy}
    // (elements are copied to scalars first: __builtin_bit_cast applied directly to `r.y` reads element 0 with this compiler)
    n2v_u32x2 r = __builtin_amdgcn_permlane16_swap(__builtin_bit_cast(unsigned, m), __builtin_bit_cast(unsigned, m), false, false);
    unsigned lo = r.x, hi = r.y;
    m = __builtin_bit_cast(float, lo) + __builtin_bit_cast(float, hi);
    r = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(unsigned, m), __builtin_bit_cast(unsigned, m), false, false);
    lo = r.x; hi = r.y;
    return __builtin_bit_cast(float, lo) + __builtin_bit_cast(float, hi);
}

template <int VEC, int NV>
struct NegSet {
    int32_t tv;                 // lanes 0..4: the five targets (lane form, for the any-match test)
    int32_t cnt;                // lanes 0..4: their token counts (instantiations that handle hot rows: fetched with the rows)
    STALE(unsigned int ver;)    // lanes 0..4: the targets' update counters when their rows were requested
    int32_t tgt[SGNS_NEG];
    float y[SGNS_NEG][NV][VEC]; // their SynNeg rows (in flight, then updated in place)
};

// FULL: d == NV * VEC * 64, rows need no tail guard.  ALLC: R >= window, every context row is in the LDS window.
// With both, the pair loop has a STATIC number of memory operations per pair (skipped targets and exhausted prefetch slots
// go to a per-wave dummy row instead of branching), which is what lets the compiler keep two pairs' rows in flight with
// counted s_waitcnt vmcnt(N) instead of draining to vmcnt(0) at every control-flow merge.
// PF: pairs whose negative rows are requested ahead (2 by default; 1 keeps 10 instead of 15 rows of a wavefront open between load and store).
// RELOAD (Hogwild launches, default): the update of a negative row is applied to the row AS IT IS NOW -- the five rows are fetched again
// right after the dot products (the gradient still uses the copy requested PF pairs ahead) and leave as `row_now + g * xc` -- and the centre's
// positive row leaves as an atomic add of what this centre changed.  A store another wavefront makes between a row's first load and its
// store is no longer overwritten: the window in which it can be lost shrinks from (PF + 1) pair steps to one reload round trip.  The CPU
// replay of this kernel's concurrency (scripts/hogwild_emul) attributes ~90 % of Hogwild's MAP loss to those overwritten negative-row
// updates and the rest to the centre row's; stale gradients themselves cost nothing.
// HOT ROWS (power-law graphs; !ALLC instantiations with A.hot_thr > 0).  A node that makes up the fraction p of all tokens sits in
// W x (2R+1) x p LDS windows at once; every one of those copies trains for ~2R+1 centres on a stale base and leaves as a delta -- for a
// hub that is hundreds of concurrent copies whose deltas ADD UP (measured on R-MAT scale 17, 985 wavefronts: MAP -15 % against the sequential
// algorithm, with or without RELOAD; the SBM graphs have no such node).  Nodes whose expected number of concurrent copies reaches 1
// (counts[v] >= tokens / (W x (2R+1))) therefore never enter the window: as a context their row is fetched for the pair and takes its
// neu1e by atomic add (RELOAD) or a plain store, like the contexts beyond the cached radius.
// PART (sgns_part.hip): the same walk-ordered TrainModel restricted to ONE bucket of the partitioned N-GPU schedule -- contexts of partition
// A.ctx_part, centre words (and negatives) of partition A.word_part, rows addressed by their local index v / parts.  Tokens are tagged once per walk
// (lane-parallel, in LDS): bits 0..28 local row, bit 29 "a centre of this bucket", bit 30 "a context of this bucket"; a token of any other partition is
// -1 and costs its centre step the window bookkeeping only.  Every rank scans every walk of an episode once per round and trains its bucket of it:
// what crosses the fabric is walks (4 bytes per token), not pairs (8 bytes x ~10 per token), and the pair order inside a bucket is TrainModel's.
template <int VEC, int NV, bool DELTA, bool FULL, bool ALLC, int PF = 2, bool RELOAD = false, bool PART = false>
__global__ __launch_bounds__(64) void sgns_win_kernel(SgnsArgs A)
{
    static_assert(PF == 1 || PF == 2, "prefetch distance");
    static_assert(!RELOAD || DELTA, "RELOAD is a Hogwild (delta write-back) mode");
    static_assert(PF == 2 || (FULL && ALLC), "the shorter prefetch exists for the all-cached full-row kernel only");
    extern __shared__ __attribute__((aligned(16))) int32_t lds[];
    constexpr int RW = NV * VEC * WAVE;              // floats per cached row (row padded to the wave's footprint)
    constexpr int NS = 2;                            // negative samples per lane and centre: 2*window*5 <= 128
    const int lane = lane_id();
    const int64_t gw = blockIdx.x;
    if (gw >= A.nwaves) return;
    // (measured and dropped, round 5: d as a compile-time constant in the FULL instantiations -- every row address a shift instead of a 64-bit scalar multiply,
    // 160 -> 66 s_mul_i32 in the ISA, but 169 -> 196 SGPR spills: 9.81 against 9.78 s per pass at SBM 1M/10M, 10.53 against 10.46 s on R-MAT scale 22 with
    // three walks per node, libraries alternated: profiles/r05_ab_sgns_const_d_dropped.jsonl)
    const int d = A.d, win = A.window, len = A.walk_len, R = A.cache_radius, S = 2 * R + 1;
    const int nsamp = 2 * win * SGNS_NEG;
    int32_t *tok = lds;
    constexpr int32_t PT_ROW = (1 << 29) - 1, PT_WORD = 1 << 29, PT_CTX = 1 << 30;
    // the token at position k as a centre word / as a context: its (local) row, or -1 when it is none in this launch (padding; PART: another partition)
    auto tok_w = [&](int k) -> int32_t { const int32_t v = tok[k]; if constexpr (PART) return (v >= 0 && (v & PT_WORD)) ? (v & PT_ROW) : -1; else return v; };
    auto tok_c = [&](int k) -> int32_t { const int32_t v = tok[k]; if constexpr (PART) return (v >= 0 && (v & PT_CTX)) ? (v & PT_ROW) : -1; else return v; };
    int32_t *negs = tok + len;                       // [2][nsamp]
    float *rowsL = reinterpret_cast<float *>(lds + ((len + 2 * nsamp + 3) & ~3));
    // slot S of rowsL stages a context row that is not cached -- instantiations that cache every context (ALLC) do not carry it: at d = 128, R = 10 that
    // brings a wavefront's LDS from 23 136 to 22 624 bytes, i.e. from six to SEVEN wavefronts per CU (plan_sgns_launch sizes the launch the same way)
    // DELTA only: the rows as loaded.  In LDS next to the working copies -- except in the bucket kernels (PART), which keep them in a per-wavefront
    // global scratch (written once when a row enters, read once when it leaves; it stays in L2): half the LDS per wavefront lets 12 instead of 7 share a CU,
    // and a bucket launch, whose wavefronts wait on row fetches most of the time (section 6 of DESIGN.md), needs the wavefronts more than the LDS
    float *rowsO = PART ? A.scratch + (size_t)gw * (size_t)(2 * R + 1) * RW : rowsL + (size_t)(S + (ALLC ? 0 : 1)) * RW;

    auto lds_ld = [&](const float *row, float (&v)[NV][VEC]) {
#pragma unroll
        for (int c = 0; c < NV; ++c)
#pragma unroll
            for (int k = 0; k < VEC; ++k) v[c][k] = row[(c * WAVE + lane) * VEC + k];
    };
    auto lds_st = [&](float *row, const float (&v)[NV][VEC]) {
#pragma unroll
        for (int c = 0; c < NV; ++c)
#pragma unroll
            for (int k = 0; k < VEC; ++k) row[(c * WAVE + lane) * VEC + k] = v[c][k];
    };
    const int dg = FULL ? NV * VEC * WAVE : d;       // guard bound of ld_row/st_row: a compile-time constant when FULL
    auto g_ld = [&](const float *p, float (&v)[NV][VEC]) {
#pragma unroll
        for (int c = 0; c < NV; ++c) ld_row<VEC>(p, dg, lane, c, v[c]);
    };
    auto g_st = [&](float *p, const float (&v)[NV][VEC]) {
#pragma unroll
        for (int c = 0; c < NV; ++c) st_row<VEC>(p, dg, lane, c, v[c]);
    };

    float *dummy = A.dummy + (size_t)gw * RW;        // this wave's private sink / source for predicated-off row traffic
    auto is_hot = [&](int32_t v) -> bool {           // (wave-uniform v: a scalar load; PART: v is a local row of the context partition)
        if constexpr (ALLC) return false;
        else return A.hot_thr > 0 && A.counts[PART ? (int64_t)v * A.parts + A.ctx_part : (int64_t)v] >= A.hot_thr;
    };
    auto o_st = [&](int slot, const float (&v)[NV][VEC]) {            // the row as loaded (delta write-back)
        if constexpr (PART) {
#pragma unroll
            for (int c = 0; c < NV; ++c) st_row<VEC>(rowsO + (size_t)slot * RW, NV * VEC * WAVE, lane, c, v[c]);
        } else lds_st(rowsO + (size_t)slot * RW, v);
    };
    auto o_ld = [&](int slot, float (&v)[NV][VEC]) {
        if constexpr (PART) {
#pragma unroll
            for (int c = 0; c < NV; ++c) ld_row<VEC>(rowsO + (size_t)slot * RW, NV * VEC * WAVE, lane, c, v[c]);
        } else lds_ld(rowsO + (size_t)slot * RW, v);
    };
    unsigned long long npairs = 0;
    PROF_DECL;
    PROF_START();
    for (int64_t wl = A.walk_lo + gw; wl < A.walk_hi; wl += A.nwaves) {
        const int32_t *walk = A.walks + wl * len;
        int64_t wid = A.walk_id_offset + wl;
        if constexpr (PART) {
            if (A.seg) {
                const int64_t sg = wl / A.seg_len, j = wl - sg * A.seg_len;
                if (j >= A.seg[A.nseg + sg]) continue;
                walk = A.walks + (A.seg[sg] + j) * len;
                wid = A.seg[2 * A.nseg + sg] + j;
            }
            for (int k = lane; k < len; k += WAVE) {
                const int32_t v = walk[k];
                int32_t t = -1;
                if (v >= 0) {
                    const int32_t pv = v % A.parts;
                    const int32_t tag = (pv == A.word_part ? PT_WORD : 0) | (pv == A.ctx_part ? PT_CTX : 0);
                    t = tag ? ((v / A.parts) | tag) : -1;
                }
                tok[k] = t;
            }
        } else {
            for (int k = lane; k < len; k += WAVE) tok[k] = walk[k];
        }
        __builtin_amdgcn_wave_barrier();
        const uint32_t w_lo = (uint32_t)wid, w_hi = (uint32_t)((uint64_t)wid >> 32);

        // slot directory: lane s < S describes slot s
        int32_t slot_node = -1, slot_ref = 0;

        // --- negative-target pipeline: stage A for centre p (Philox draws + one gather of {X, UT[X], KT[X]} per sample, A.SK), finalize -> LDS one centre later
        int32_t XA[NS], KA[NS]; float uA[NS], UA[NS];
        // Only the samples of contexts the centre will train on are drawn: the window shrink b of centre p is itself a Philox draw, the
        // draws are counter-based (skipping one changes no other), and the uncoalesced table gathers per centre turned out to be what
        // caps the kernel (scripts/microbench/rows.hip "mix": 6.2 -> 4.5 G rows/s with them) -- 45 % of the slots are never used.
        auto stage_a = [&](int p) {
            int bp = 0;
            if constexpr (PART) { if (p < len && __builtin_amdgcn_readfirstlane(tok_w(p)) < 0) p = len; }      // not a centre of this bucket: nothing is drawn
            if (p < len) {
                const u32x4 rwp = philox4x32_10(A.seed, w_lo, w_hi, (uint32_t)p, (uint32_t)TAG_WIN | ((uint32_t)A.epoch << 8));
                bp = (int)(rwp.x % (uint32_t)win);
            }
#pragma unroll
            for (int k = 0; k < NS; ++k) {
                const int s = lane + k * WAVE;
                XA[k] = 0; uA[k] = 0.f; UA[k] = 2.f; KA[k] = 0;             // (a sample that is not drawn finalizes to target 0, which nothing reads)
                if (p < len && s < nsamp) {
                    const int ai = s / SGNS_NEG;
                    const int a = ai < win ? ai : ai + 1;
                    const int cp = p - win + a;
                    if (a >= bp && a < 2 * win + 1 - bp && cp >= 0 && cp < len) {
                        const int j = s - ai * SGNS_NEG + 1;
                        const u32x4 rn = philox4x32_10(A.seed, w_lo, w_hi, (uint32_t)p | ((uint32_t)a << 16),
                                                       (uint32_t)TAG_NEG | ((uint32_t)A.epoch << 8) | ((uint32_t)j << 16));
                        const uint4 e = A.SK[mulhi_range(rn.x, A.n)];          // RndUnigramInt (ELF @0x40d5f0): slot -> {X, UTable[X], KTable[X]}, one 16-byte gather
                        XA[k] = (int32_t)e.x; UA[k] = __builtin_bit_cast(float, e.y); KA[k] = (int32_t)e.z;
                        uA[k] = u01(rn.y);
                    }
                }
            }
        };
        // ... and the "special" mask of centre p: bit ai is set when the (centre, context) pair of slot ai cannot take the
        // all-targets-independent fast path -- a target equals the centre word (TrainModel skips it), a target was drawn twice
        // (the second use must see the first update), or a target also occurs in one of the previous two slots (this slot's rows
        // were requested before those slots' updates were stored, so the slow path fetches them again).  One lane per slot,
        // once per centre; at n = 1M it is set for ~1e-4 of the pairs, on karate (n = 34) for nearly all.
        auto stage_fin = [&](int p, int buf) -> uint32_t {
            int32_t *dst = negs + buf * nsamp;
#pragma unroll
            for (int k = 0; k < NS; ++k) {
                const int s = lane + k * WAVE;
                if (s < nsamp) dst[s] = (uA[k] < UA[k]) ? XA[k] : KA[k];
            }
            if (p >= len) return 0u;
            const int32_t wordn = __builtin_amdgcn_readfirstlane(tok_w(p));
            bool sp = false;
            if (lane < 2 * win) {
                int32_t t[SGNS_NEG];
#pragma unroll
                for (int j = 0; j < SGNS_NEG; ++j) { t[j] = dst[lane * SGNS_NEG + j]; sp = sp || t[j] == wordn; }
#pragma unroll
                for (int j = 0; j < SGNS_NEG; ++j)
#pragma unroll
                    for (int jp = 0; jp < j; ++jp) sp = sp || t[j] == t[jp];
                // (PART: the two pairs in flight ahead of a slot are the previous two slots WITH a context of this bucket, known only once the
                // centre's context mask is -- the centre adds that test itself, see `cross` below)
                if constexpr (!PART) {
#pragma unroll
                    for (int k = 0; k < 2 * SGNS_NEG; ++k) {
                        const int idx = (lane - 2) * SGNS_NEG + k;
                        const int32_t u = dst[idx >= 0 ? idx : 0];
#pragma unroll
                        for (int j = 0; j < SGNS_NEG; ++j) sp = sp || (idx >= 0 && t[j] == u);
                    }
                }
            }
            return (uint32_t)__builtin_amdgcn_ballot_w64(sp);
        };
        // PART, WHOLE-WALK mode: a bucket's contexts are the 1 / parts of the tokens that belong to its SynPos partition.  When all of them fit the
        // window's slots (always from ~4 partitions on) they enter TOGETHER before the first centre -- four rows in flight at a time instead of one
        // exposed round trip per token -- and leave together after the last centre; the centre loop then does no window bookkeeping at all and visits
        // ONLY the positions that hold a centre word of this bucket (the other 1 - 1 / parts of a walk cost ~0.5 us each as empty iterations: most
        // of a bucket launch's time at 8 partitions).  One wavefront: the window is transparent, so the tables are the same either way.
        bool whole = false;
        unsigned long long actm0 = 0ull, actm1 = 0ull;          // positions 0..63 / 64..127 that hold a centre word of this bucket
        if constexpr (PART) {
            int nctx = 0;
            for (int base = 0; base < len; base += WAVE)
                nctx += (int)__builtin_popcountll(__builtin_amdgcn_ballot_w64(base + lane < len && tok_c(base + lane < len ? base + lane : 0) >= 0));
            whole = nctx <= S && len <= 2 * WAVE;
            if (whole) {
                actm0 = __builtin_amdgcn_ballot_w64(lane < len && tok_w(lane < len ? lane : 0) >= 0);
                actm1 = __builtin_amdgcn_ballot_w64(WAVE + lane < len && tok_w(WAVE + lane < len ? WAVE + lane : 0) >= 0);
                if (!(actm0 | actm1)) { __builtin_amdgcn_wave_barrier(); continue; }        // no centre word of this bucket in the walk: nothing to train
            }
        }
        auto next_act = [&](int p) -> int {                     // first position >= p that holds a centre word of this bucket (len if none)
            if (p < WAVE) { const unsigned long long m = actm0 >> p; if (m) return p + (int)__builtin_ctzll(m); p = WAVE; }
            if (p < 2 * WAVE) { const unsigned long long m = actm1 >> (p - WAVE); if (m) return p + (int)__builtin_ctzll(m); }
            return len;
        };
        int pos_first = 0;
        uint32_t spec_next;
        if constexpr (PART) {
            pos_first = whole ? next_act(0) : 0;
            stage_a(pos_first);
            spec_next = stage_fin(pos_first, 0);
        } else {
            stage_a(0);
            spec_next = stage_fin(0, 0);
        }

        // --- cache: enter token q (directory now, row through `rowE` -> LDS by the caller), leave token q
        // rows of tokens 0 .. R-1 (whole-walk mode: of every token) enter before the first centre
        for (int q = 0; (PART && whole) ? q < len : (q < R && q < len); ++q) {
            const int32_t v = __builtin_amdgcn_readfirstlane(tok_c(q));
            if (v < 0 || is_hot(v)) continue;
            const unsigned long long hit = __builtin_amdgcn_ballot_w64(slot_node == v);
            if (hit) { if (lane == (int)__builtin_ctzll(hit)) ++slot_ref; continue; }
            const int s = (int)__builtin_ctzll(__builtin_amdgcn_ballot_w64(slot_node < 0 && lane < S));
            if (!(PART && whole)) {
                float r[NV][VEC];
                g_ld(A.SynPos + (int64_t)v * d, r);
                lds_st(rowsL + (size_t)s * RW, r);
                if constexpr (DELTA) o_st(s, r);
            }
            if (lane == s) { slot_node = v; slot_ref = 1; }
        }
        if constexpr (PART) {
            if (whole) {
                unsigned long long occ = __builtin_amdgcn_ballot_w64(slot_node >= 0 && lane < S);
                while (occ) {
                    int sl[4]; float r4[4][NV][VEC];
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        sl[u] = occ ? (int)__builtin_ctzll(occ) : -1;
                        if (occ) occ &= occ - 1;
                        if (sl[u] >= 0) g_ld(A.SynPos + (int64_t)__builtin_amdgcn_readlane(slot_node, sl[u]) * d, r4[u]);
                    }
#pragma unroll
                    for (int u = 0; u < 4; ++u)
                        if (sl[u] >= 0) {
                            lds_st(rowsL + (size_t)sl[u] * RW, r4[u]);
                            if constexpr (DELTA) o_st(sl[u], r4[u]);
                        }
                }
            }
        }

        // data and address registers of the stores a centre ends with.  They are kept alive (empty asm "uses" inside the next centre's pair steps) so
        // that the register allocator cannot hand them out again right away: overwriting the source registers of a store that is still in flight is
        // a write-after-read hazard the compiler guards with s_waitcnt vmcnt(0) -- a full drain that waits for the store's acknowledgement (seen in the
        // ISA: three such drains per centre, ~a quarter of the kernel's time)
        float tail_d0[NV][VEC], tail_d1[NV][VEC]; float *tail_p0[NV], *tail_p1[NV];
#pragma unroll
        for (int c = 0; c < NV; ++c) {
            tail_p0[c] = tail_p1[c] = nullptr;
#pragma unroll
            for (int k = 0; k < VEC; ++k) tail_d0[c][k] = tail_d1[c][k] = 0.f;
        }
        auto tail_keep = [&]() __attribute__((always_inline)) {
#pragma unroll
            for (int c = 0; c < NV; ++c) {
                asm volatile("" ::"v"(tail_p0[c]), "v"(tail_p1[c]));
#pragma unroll
                for (int k = 0; k < VEC; ++k) asm volatile("" ::"v"(tail_d0[c][k]), "v"(tail_d1[c][k]));
            }
        };
        int pos_next = 0, par = 1;                              // (par: PART whole-walk mode -- parity of the centre's ordinal among the visited ones; toggled at the top of an iteration)
        for (int pos = PART ? pos_first : 0; pos < len; pos = PART ? pos_next : pos + 1) {
            // the next position the loop will visit (whole-walk mode: the next one that holds a centre word of this bucket)
            int nx1 = pos + 1;
            if constexpr (PART) {
                if (whole) nx1 = next_act(pos + 1);
                pos_next = nx1; par ^= 1;
            }
            const int32_t word = __builtin_amdgcn_readfirstlane(tok_w(pos));
            uint32_t spec_cur = spec_next;
            bool fin_done = false;
            // token pos+R enters
            float rowE[NV][VEC]; int sE = -1;
            if (pos + R < len && !(PART && whole)) {
                const int32_t v = __builtin_amdgcn_readfirstlane(tok_c(pos + R));
                if (v >= 0 && !is_hot(v)) {
                    const unsigned long long hit = __builtin_amdgcn_ballot_w64(slot_node == v);
                    if (hit) { if (lane == (int)__builtin_ctzll(hit)) ++slot_ref; }
                    else {
                        sE = (int)__builtin_ctzll(__builtin_amdgcn_ballot_w64(slot_node < 0 && lane < S));
                        g_ld(A.SynPos + (int64_t)v * d, rowE);
                        if (lane == sE) { slot_node = v; slot_ref = 1; }
                    }
                }
            }
            // token pos-R leaves after this centre: when it is the last holder of its slot, fetch the row as it is NOW
            float rowG[NV][VEC], rowO[NV][VEC]; int sX = -1; int32_t vX = -1;
            if (pos - R >= 0 && !(PART && whole)) {
                vX = __builtin_amdgcn_readfirstlane(tok_c(pos - R));
                const unsigned long long hx = vX >= 0 ? __builtin_amdgcn_ballot_w64(slot_node == vX) : 0ull;     // (no slot: a hot row, never cached)
                if (hx) {
                    const int s = (int)__builtin_ctzll(hx);
                    const int refc = __builtin_amdgcn_readlane(slot_ref, s) - 1;
                    if (lane == s) slot_ref = refc;
                    if (refc == 0) {
                        sX = s;
                        if constexpr (DELTA && !RELOAD) g_ld(A.SynPos + (int64_t)vX * d, rowG);      // (RELOAD: the row leaves as an atomic add of its change)
                    }
                }
            }
            // negatives of the next centre the loop will visit: drawn and gathered now, finalized inside this centre's first pair step
            PROF_LAP(0);
            if constexpr (PART) stage_a(nx1); else stage_a(pos + 1);
            PROF_LAP(5);                                             // negative-target pipeline: Philox + table gather

            float yp[NV][VEC], yp0[NV][VEC];                         // the centre's positive row SynNeg[word] (and, RELOAD, as it was loaded)
            float *pp = A.SynNeg + (int64_t)(word >= 0 ? word : 0) * d;
            // FRESH HOT ROWS, bit 0: the centre word is a hot row -> its positive row is refreshed by every pair's returning atomic add (pos_update)
            bool word_hot = false;
            if constexpr (!ALLC && RELOAD) {
                if (word >= 0 && A.hot_thr > 0) {      // a LOCALLY hot word (hotkey INT32_MAX: n2v.hip ensure_hotkey) always; a count-hot one under fresh bit 0
                    const int32_t cw = A.counts[PART ? (int64_t)word * A.parts + A.word_part : (int64_t)word];
                    word_hot = cw == INT32_MAX || ((A.fresh & 1) && cw >= A.hot_thr);
                }
            }
            STALE(unsigned int ver_c = 0u; int cnt_c = 1;)
            if (word >= 0) {
                const int64_t t = A.token_offset + wl * len + pos;
                const int64_t tq = t - (t % 10000);
                float alpha = A.alpha0 * (1.0f - (float)((double)tq / (double)A.denom));
                alpha = fmaxf(alpha, A.alpha0 * 0.0001f);
                const u32x4 rw = philox4x32_10(A.seed, w_lo, w_hi, (uint32_t)pos, (uint32_t)TAG_WIN | ((uint32_t)A.epoch << 8));
                const int b = (int)(rw.x % (uint32_t)win);
                const int32_t *ncur = negs + ((PART && whole) ? par : (pos & 1)) * nsamp;

                g_ld(pp, yp);
                STALE(if constexpr (!PART) { ver_c = stale_load(A.stale_ver + word); cnt_c = A.counts[word]; })
                if constexpr (RELOAD) {
#pragma unroll
                    for (int c = 0; c < NV; ++c)
#pragma unroll
                        for (int k = 0; k < VEC; ++k) yp0[c][k] = yp[c][k];
                }

                // the contexts of this centre, ascending (TrainModel's order): bit a <-> position pos - win + a
                bool valid = false;
                {
                    const int a = lane, cp = pos - win + a;
                    if (a >= b && a < 2 * win + 1 - b && a != win && cp >= 0 && cp < len) valid = tok_c(cp) >= 0;
                }
                unsigned long long m_proc = __builtin_amdgcn_ballot_w64(valid), m_iss = m_proc;
                npairs += (unsigned long long)__builtin_popcountll(m_proc);
                {   // the slots after ai in flight are ai+1, ai+2 only if the valid contexts are contiguous (always, but for padded walks)
                    const unsigned long long lowm = (1ull << win) - 1ull;
                    const unsigned long long mai = (m_proc & lowm) | ((m_proc >> (win + 1)) << win);
                    const unsigned long long sh = mai ? (mai >> __builtin_ctzll(mai)) : 0ull;
                    if constexpr (!PART) { if (sh & (sh + 1ull)) spec_cur = 0xFFFFFFFFu; }
                    else {
                        // the contexts of this bucket are scattered over the window: slot ai's rows were requested before the updates of the previous two
                        // slots THAT HAVE A CONTEXT were stored.  Lane ai compares its five targets with theirs (once per centre, lane-parallel).
                        bool cross = false;
                        if (lane < 2 * win && ((mai >> lane) & 1ull)) {
                            unsigned long long below = mai & ((1ull << lane) - 1ull);
                            int32_t t[SGNS_NEG];
#pragma unroll
                            for (int j = 0; j < SGNS_NEG; ++j) t[j] = ncur[lane * SGNS_NEG + j];
#pragma unroll
                            for (int h = 0; h < 2; ++h) {
                                if (below) {
                                    const int pa = 63 - __builtin_clzll(below);
                                    below &= ~(1ull << pa);
#pragma unroll
                                    for (int k = 0; k < SGNS_NEG; ++k) {
                                        const int32_t u = ncur[pa * SGNS_NEG + k];
#pragma unroll
                                        for (int j = 0; j < SGNS_NEG; ++j) cross = cross || t[j] == u;
                                    }
                                }
                            }
                        }
                        spec_cur |= (uint32_t)__builtin_amdgcn_ballot_w64(cross);
                    }
                }

                NegSet<VEC, NV> q0, q1, q2;          // three register sets rotate: processed now / next / the one after
                auto issue = [&](NegSet<VEC, NV> &Q) __attribute__((always_inline)) {
                    const bool live = m_iss != 0;                   // exhausted: the same five loads, from the dummy row
                    const int a = live ? (int)__builtin_ctzll(m_iss) : 0;
                    m_iss &= m_iss - 1;
                    const int ai = a < win ? a : a - 1;
                    Q.tv = ncur[ai * SGNS_NEG + (lane < SGNS_NEG ? lane : 0)];
                    if (lane >= SGNS_NEG || !live) Q.tv = -1;
                    if constexpr (!ALLC && RELOAD) Q.cnt = A.counts[Q.tv >= 0 ? (PART ? (int64_t)Q.tv * A.parts + A.word_part : (int64_t)Q.tv) : 0];      // one 4-byte gather per pair, in flight with the rows
                    STALE(if constexpr (!PART) { if constexpr (ALLC || !RELOAD) Q.cnt = A.counts[Q.tv >= 0 ? Q.tv : 0]; Q.ver = stale_load(A.stale_ver + (Q.tv >= 0 ? Q.tv : 0)); })
#pragma unroll
                    for (int j = 0; j < SGNS_NEG; ++j) {
                        Q.tgt[j] = __builtin_amdgcn_readlane(Q.tv, j);
                        g_ld(live ? A.SynNeg + (int64_t)Q.tgt[j] * d : dummy, Q.y[j]);
                    }
                };
                issue(q0);
                if constexpr (PF == 2) issue(q1);

                if (sE >= 0) {               // the entering row has landed by now (requested before everything above)
                    lds_st(rowsL + (size_t)sE * RW, rowE);
                    if constexpr (DELTA) o_st(sE, rowE);
                }
                // the centre row is needed by the first pair anyway; "using" it here pins its wait BEFORE the pair loop (a counted wait: the prefetches just
                // issued stay in flight), so that the compiler does not have to drain everything when it meets yp again after a loop it cannot count through
#pragma unroll
                for (int c = 0; c < NV; ++c)
#pragma unroll
                    for (int k = 0; k < VEC; ++k) asm volatile("" ::"v"(yp[c][k]));
                PROF_LAP(6);                                         // centre set-up: alpha, window draw, masks, centre row request, first prefetches, entering row -> LDS

                // the positive target's update, gradient scale g0, context row xc: neu1e += g0 * yp (the copy the gradient was computed from), then
                // yp += g0 * xc -- in registers, or (hot centre word, FRESH bit 0) as a returning atomic add whose result is the row as memory holds it now
                auto pos_update = [&](float g0, const float (&xc)[NV][VEC], float (&neu)[NV][VEC]) __attribute__((always_inline)) {
                    bool atomic_path = false;
                    if constexpr (!ALLC && RELOAD) atomic_path = word_hot;
                    if (atomic_path) {
#pragma unroll
                        for (int c = 0; c < NV; ++c)
#pragma unroll
                            for (int k = 0; k < VEC; ++k) {
                                neu[c][k] = fmaf(g0, yp[c][k], neu[c][k]);
                                const float dl = g0 * xc[c][k];
                                float old = 0.f;
                                if ((c * WAVE + lane) * VEC + k < dg)
                                    old = __builtin_amdgcn_global_atomic_fadd_f32((__attribute__((address_space(1))) float *)(pp + (c * WAVE + lane) * VEC + k), dl);
                                yp[c][k] = (c * WAVE + lane) * VEC + k < dg ? old + dl : 0.f;
                                yp0[c][k] = yp[c][k];
                            }
                    } else {
#pragma unroll
                        for (int c = 0; c < NV; ++c)
#pragma unroll
                            for (int k = 0; k < VEC; ++k) {
                                neu[c][k] = fmaf(g0, yp[c][k], neu[c][k]);
                                yp[c][k] = fmaf(g0, xc[c][k], yp[c][k]);
                            }
                    }
#ifdef GEMHIP_SGNS_STALENESS
                    if constexpr (!PART) {
                        if (lane == 0) {
                            if (atomic_path) { const unsigned int v = atomicAdd(A.stale_ver + word, 1u); stale_rec(A.stale_hist, 1, cnt_c, v - ver_c); ver_c = v + 1u; }
                            else stale_rec(A.stale_hist, 1, cnt_c, stale_load(A.stale_ver + word) - ver_c);
                        }
                    }
#endif
                };
                // one (centre, context) pair: C holds its negative rows, the sets in between are in flight, P2 is free
                auto step = [&](NegSet<VEC, NV> &C, NegSet<VEC, NV> &P2) __attribute__((always_inline)) {
                    PROF_LAP(0);                                     // outside the pair steps (per-centre work, loop control)
                    const int a = (int)__builtin_ctzll(m_proc);
                    m_proc &= m_proc - 1;
                    issue(P2);
                    PROF_LAP(1);                                     // issue of the prefetch

                    const int32_t ctx = __builtin_amdgcn_readfirstlane(tok_c(pos - win + a));
                    const unsigned long long chit = __builtin_amdgcn_ballot_w64(slot_node == ctx);
                    float *lrow = rowsL + (size_t)((ALLC || chit) ? (int)__builtin_ctzll(chit) : S) * RW;
                    float *pc = A.SynPos + (int64_t)ctx * d;
                    float xc[NV][VEC], neu[NV][VEC];
                    STALE(unsigned int ver_x = 0u;)
                    if constexpr (!ALLC)
                        if (!chit) {        // beyond the cached radius: stage through LDS so that the wait for this row stays inside the branch
                            float t[NV][VEC];
                            g_ld(pc, t);
                            STALE(if constexpr (!PART) ver_x = stale_load(A.stale_ver + A.n_nodes + ctx);)
                            lds_st(lrow, t);
                        }
                    lds_ld(lrow, xc);
#pragma unroll
                    for (int c = 0; c < NV; ++c)
#pragma unroll
                        for (int k = 0; k < VEC; ++k) neu[c][k] = 0.f;
                    const int ai_c = a < win ? a : a - 1;
                    PROF_LAP(2);                                     // context lookup + LDS read (includes its lgkmcnt wait)
                    if constexpr (PF == 2) PROF_WAIT_VM(10); else PROF_WAIT_VM(5);
                    PROF_LAP(3);                                     // waiting for this pair's rows (the younger prefetches may stay in flight)
                    tail_keep();                                     // (no instruction: the previous centre's store registers stay reserved up to here)
                    if (!fin_done) {
                        // the negative targets of the NEXT centre: their table gather (stage_a, issued before this centre's first prefetch) is older than
                        // the rows just waited for, so consuming it HERE costs no wait; after the pair loop it would be a full drain
                        if constexpr (PART) spec_next = stage_fin(nx1, whole ? (par ^ 1) : (nx1 & 1));
                        else spec_next = stage_fin(pos + 1, (pos + 1) & 1);
                        fin_done = true;
                    }
                    if (!((spec_cur >> ai_c) & 1u)) {
                        // fast path: the six targets are distinct rows and none is the centre word -> six independent updates
                        // hot negative rows (hubs drawn as negatives by many wavefronts at once): their update is an atomic add of g * xc -- no
                        // window at all -- and the row-sized store goes to the scratch row instead (the number of loads / stores per pair stays static)
                        unsigned hotm = 0u;
                        if constexpr (!ALLC && RELOAD) {
                            hotm = A.hot_thr > 0 ? (unsigned)__builtin_amdgcn_ballot_w64(lane < SGNS_NEG && C.cnt >= A.hot_thr) : 0u;
                            if ((A.fresh & 4) && A.hot_thr > 0) hotm |= (unsigned)__builtin_amdgcn_ballot_w64(lane < SGNS_NEG && C.cnt >= A.neg_thr);     // bit 2: see SgnsArgs::fresh
                            if ((A.fresh & 2) && hotm) {         // FRESH bit 1: the gradient of a hot row is computed from the row as it is NOW
#pragma unroll
                                for (int j = 0; j < SGNS_NEG; ++j)
                                    if ((hotm >> j) & 1u) g_ld(A.SynNeg + (int64_t)C.tgt[j] * d, C.y[j]);
                                STALE(if constexpr (!PART) { if (lane < SGNS_NEG && ((hotm >> lane) & 1u)) C.ver = stale_load(A.stale_ver + C.tv); })
                            }
                        }
                        float part[6];
#pragma unroll
                        for (int j = 0; j < 6; ++j) part[j] = 0.f;
#pragma unroll
                        for (int c = 0; c < NV; ++c)
#pragma unroll
                            for (int k = 0; k < VEC; ++k) {
                                part[0] = fmaf(xc[c][k], yp[c][k], part[0]);
#pragma unroll
                                for (int j = 0; j < SGNS_NEG; ++j) part[j + 1] = fmaf(xc[c][k], C.y[j][c][k], part[j + 1]);
                            }
                        float Rn[RELOAD ? SGNS_NEG : 1][NV][VEC];
                        if constexpr (RELOAD) {      // the five rows as they are NOW (requested here, needed after the sigmoid)
#pragma unroll
                            for (int j = 0; j < SGNS_NEG; ++j) g_ld(A.SynNeg + (int64_t)C.tgt[j] * d, Rn[j]);
                        }
                        const float f = wave_sum6(part, lane);
                        const float gl = sgns_grad_fast(f, (lane & 7) == 0 ? 1.0f : 0.0f, alpha);     // lanes 0..5: g of target 0..5
                        float g[6];
#pragma unroll
                        for (int j = 0; j < 6; ++j) g[j] = bcast_lane(gl, j);
                        pos_update(g[0], xc, neu);
#ifdef GEMHIP_SGNS_STALENESS
                        if constexpr (!PART) {
                            if (lane < SGNS_NEG && C.tv >= 0) { const unsigned int v = atomicAdd(A.stale_ver + C.tv, 1u); stale_rec(A.stale_hist, 0, C.cnt, v - C.ver); }
                        }
#endif
                        if constexpr (RELOAD) {
#pragma unroll
                            for (int j = 0; j < SGNS_NEG; ++j)
#pragma unroll
                                for (int c = 0; c < NV; ++c)
#pragma unroll
                                    for (int k = 0; k < VEC; ++k) neu[c][k] = fmaf(g[j + 1], C.y[j][c][k], neu[c][k]);
#pragma unroll
                            for (int j = 0; j < SGNS_NEG; ++j) {
                                const bool hotj = (hotm >> j) & 1u;
                                float *pj = A.SynNeg + (int64_t)C.tgt[j] * d;
                                if (hotj && g[j + 1] != 0.f) {          // (sigma clamped to 0 beyond -MaxExp: the update is exactly zero -- 128 atomic adds of 0.0 saved, nothing changed)
#pragma unroll
                                    for (int c = 0; c < NV; ++c)
#pragma unroll
                                        for (int k = 0; k < VEC; ++k)
                                            if ((c * WAVE + lane) * VEC + k < dg)
                                                __builtin_amdgcn_global_atomic_fadd_f32((__attribute__((address_space(1))) float *)(pj + (c * WAVE + lane) * VEC + k), g[j + 1] * xc[c][k]);
                                }
#pragma unroll
                                for (int c = 0; c < NV; ++c)
#pragma unroll
                                    for (int k = 0; k < VEC; ++k) Rn[j][c][k] = fmaf(g[j + 1], xc[c][k], Rn[j][c][k]);
                                g_st(hotj ? dummy : pj, Rn[j]);
                            }
                        } else {
#pragma unroll
                            for (int j = 0; j < SGNS_NEG; ++j) {
#pragma unroll
                                for (int c = 0; c < NV; ++c)
#pragma unroll
                                    for (int k = 0; k < VEC; ++k) {
                                        neu[c][k] = fmaf(g[j + 1], C.y[j][c][k], neu[c][k]);
                                        C.y[j][c][k] = fmaf(g[j + 1], xc[c][k], C.y[j][c][k]);
                                    }
                                g_st(A.SynNeg + (int64_t)C.tgt[j] * d, C.y[j]);
                            }
                        }
                    } else {
                        // slow path (exact sequential semantics): the rows may have been requested before an update of the same row by
                        // one of the two previous pairs was stored -- fetch them again (program order after those stores)
#pragma unroll
                        for (int j = 0; j < SGNS_NEG; ++j) g_ld(C.tgt[j] < 0 ? dummy : A.SynNeg + (int64_t)C.tgt[j] * d, C.y[j]);
                        {   // positive target (label 1), row lives in registers
                            float part = 0.f;
#pragma unroll
                            for (int c = 0; c < NV; ++c)
#pragma unroll
                                for (int k = 0; k < VEC; ++k) part = fmaf(xc[c][k], yp[c][k], part);
                            const float g = sgns_grad(wave_sum(part), 1.0f, alpha);
                            pos_update(g, xc, neu);
                        }
#pragma unroll
                        for (int j = 0; j < SGNS_NEG; ++j) {
                            const bool skip = C.tgt[j] == word || C.tgt[j] < 0;  // TrainModel: `if (Target == Word) continue` (predicated: g = 0, row -> dummy)
#pragma unroll
                            for (int jp = 0; jp < j; ++jp)                       // a target drawn twice sees the first update
                                if (C.tgt[jp] == C.tgt[j]) {
#pragma unroll
                                    for (int c = 0; c < NV; ++c)
#pragma unroll
                                        for (int k = 0; k < VEC; ++k) C.y[j][c][k] = C.y[jp][c][k];
                                }
                            float part = 0.f;
#pragma unroll
                            for (int c = 0; c < NV; ++c)
#pragma unroll
                                for (int k = 0; k < VEC; ++k) part = fmaf(xc[c][k], C.y[j][c][k], part);
                            float g = sgns_grad(wave_sum(part), 0.0f, alpha);
                            g = skip ? 0.f : g;
#pragma unroll
                            for (int c = 0; c < NV; ++c)
#pragma unroll
                                for (int k = 0; k < VEC; ++k) { neu[c][k] = fmaf(g, C.y[j][c][k], neu[c][k]); C.y[j][c][k] = fmaf(g, xc[c][k], C.y[j][c][k]); }
                            if constexpr (RELOAD) {
                                // Hogwild: the update leaves as an atomic add of g * xc, NEVER as a store of the register copy (round 6).  A pair lands here because a
                                // target repeats -- on a power-law graph that is nearly always a HUB row, which other wavefronts update by atomic add several times
                                // per microsecond: a plain store of `row as fetched at the top of this path + my update` wipes out every add that landed in
                                // between (~1 us: the re-fetch round trip plus up to six sequential dot products), and it did so for all five rows of the pair.
                                // (The register copy still carries the update forward to a later occurrence of the same target inside the pair.)
                                if (!skip) {
                                    float *pj = A.SynNeg + (int64_t)C.tgt[j] * d;
#pragma unroll
                                    for (int c = 0; c < NV; ++c)
#pragma unroll
                                        for (int k = 0; k < VEC; ++k)
                                            if ((c * WAVE + lane) * VEC + k < dg)
                                                __builtin_amdgcn_global_atomic_fadd_f32((__attribute__((address_space(1))) float *)(pj + (c * WAVE + lane) * VEC + k), g * xc[c][k]);
                                }
                            } else g_st(skip ? dummy : A.SynNeg + (int64_t)C.tgt[j] * d, C.y[j]);
                        }
                    }
#pragma unroll
                    for (int c = 0; c < NV; ++c)
#pragma unroll
                        for (int k = 0; k < VEC; ++k) xc[c][k] += neu[c][k];
                    if (ALLC || chit) lds_st(lrow, xc);
                    else if constexpr (RELOAD) {                     // uncached context row (beyond the radius, or a hot row): its neu1e by atomic add
#pragma unroll
                        for (int c = 0; c < NV; ++c)
#pragma unroll
                            for (int k = 0; k < VEC; ++k)
                                if ((c * WAVE + lane) * VEC + k < dg)
                                    __builtin_amdgcn_global_atomic_fadd_f32((__attribute__((address_space(1))) float *)(pc + (c * WAVE + lane) * VEC + k), neu[c][k]);
                        STALE(if constexpr (!PART) { if (lane == 0) { const unsigned int v = atomicAdd(A.stale_ver + A.n_nodes + ctx, 1u); stale_rec(A.stale_hist, 2, A.counts[ctx], v - ver_x); } })
                    } else g_st(pc, xc);
                    PROF_LAP(4);                                     // arithmetic + stores
                };
                if constexpr (PF == 2) {
                    while (true) {
                        if (!m_proc) break;
                        step(q0, q2);
                        if (!m_proc) break;
                        step(q1, q0);
                        if (!m_proc) break;
                        step(q2, q1);
                    }
                } else {
                    while (true) {
                        if (!m_proc) break;
                        step(q0, q1);
                        if (!m_proc) break;
                        step(q1, q0);
                    }
                }
                // (the prefetch slots filled past the last pair are dead: nothing consumes them, nothing waits for them -- round 2 "retired" them here,
                // which was a full drain including the last pair's stores)
                PROF_LAP(0);
            } else if (sE >= 0) {
                lds_st(rowsL + (size_t)sE * RW, rowE);
                if constexpr (DELTA) o_st(sE, rowE);
            }

            // End of the centre.  vmcnt counts loads and stores in order and the compiler cannot count across the pair loop, so every use of an
            // older load here is a full drain (s_waitcnt vmcnt(0)) that also waits for the acknowledgement of whatever was stored last.  Hence:
            // first everything that CONSUMES loads (the negative targets of the next centre, the leaving row as it is now) -- one drain, of
            // the last pair's stores -- and only then the centre's own stores, after which nothing waits until the next centre's first pair
            // (round 2 stored the centre row, then consumed, then stored the leaving row: three exposed round trips per centre).
            if (!fin_done) {                                        // (a centre without pairs; otherwise done inside its first pair step)
                if constexpr (PART) spec_next = stage_fin(nx1, whole ? (par ^ 1) : (nx1 & 1));
                else spec_next = stage_fin(pos + 1, (pos + 1) & 1);
            }
            if (sX >= 0) {
                float l[NV][VEC];
                lds_ld(rowsL + (size_t)sX * RW, l);
                if constexpr (DELTA) {
                    o_ld(sX, rowO);
#pragma unroll
                    for (int c = 0; c < NV; ++c)
#pragma unroll
                        for (int k = 0; k < VEC; ++k) l[c][k] = RELOAD ? l[c][k] - rowO[c][k] : rowG[c][k] + (l[c][k] - rowO[c][k]);
                }
#pragma unroll
                for (int c = 0; c < NV; ++c) {
                    tail_p1[c] = A.SynPos + (int64_t)vX * d + (c * WAVE + lane) * VEC;
#pragma unroll
                    for (int k = 0; k < VEC; ++k) tail_d1[c][k] = l[c][k];
                }
            }
            if (word >= 0) {
#pragma unroll
                for (int c = 0; c < NV; ++c) {
                    tail_p0[c] = pp + (c * WAVE + lane) * VEC;
#pragma unroll
                    for (int k = 0; k < VEC; ++k) tail_d0[c][k] = RELOAD ? yp[c][k] - yp0[c][k] : yp[c][k];
                }
            }
            // ---- stores only from here on
            STALE(if constexpr (!PART) { if (word >= 0 && !word_hot && lane == 0) atomicAdd(A.stale_ver + word, 1u); })
            if (word >= 0 && !word_hot) {           // (a hot centre word under FRESH bit 0 has already added every pair's change)
#pragma unroll
                for (int c = 0; c < NV; ++c) {
                    if ((c * WAVE + lane) * VEC < dg) {
                        if constexpr (RELOAD) {      // what this centre changed, added to the row as it is now (global_atomic_add_f32: nothing another wavefront stored is lost)
#pragma unroll
                            for (int k = 0; k < VEC; ++k) __builtin_amdgcn_global_atomic_fadd_f32((__attribute__((address_space(1))) float *)(tail_p0[c] + k), tail_d0[c][k]);
                        } else st_lane<VEC>(tail_p0[c], tail_d0[c]);
                    }
                }
            }
            if (sX >= 0) {
#pragma unroll
                for (int c = 0; c < NV; ++c)
                    if ((c * WAVE + lane) * VEC < dg) {
                        if constexpr (RELOAD) {      // the leaving window row: what this wavefront changed, added to the row as it is now
#pragma unroll
                            for (int k = 0; k < VEC; ++k) __builtin_amdgcn_global_atomic_fadd_f32((__attribute__((address_space(1))) float *)(tail_p1[c] + k), tail_d1[c][k]);
                        } else st_lane<VEC>(tail_p1[c], tail_d1[c]);
                    }
                if (lane == sX) slot_node = -1;
            }
            PROF_LAP(7);                                             // end of the centre: consume, then the centre's stores
            __builtin_amdgcn_wave_barrier();
        }
        if constexpr (PART) {
            if (whole) {        // every cached row leaves now: what this wavefront changed, as an atomic add (Hogwild) or the row itself (one wavefront)
                unsigned long long occ = __builtin_amdgcn_ballot_w64(slot_node >= 0 && lane < S);
                while (occ) {
                    const int s = (int)__builtin_ctzll(occ);
                    occ &= occ - 1;
                    float *prow = A.SynPos + (int64_t)__builtin_amdgcn_readlane(slot_node, s) * d;
                    float l[NV][VEC];
                    lds_ld(rowsL + (size_t)s * RW, l);
                    if constexpr (DELTA) {
                        float o[NV][VEC];
                        o_ld(s, o);
#pragma unroll
                        for (int c = 0; c < NV; ++c)
#pragma unroll
                            for (int k = 0; k < VEC; ++k)
                                if ((c * WAVE + lane) * VEC + k < dg)
                                    __builtin_amdgcn_global_atomic_fadd_f32((__attribute__((address_space(1))) float *)(prow + (c * WAVE + lane) * VEC + k), l[c][k] - o[c][k]);
                    } else g_st(prow, l);
                }
                slot_node = -1; slot_ref = 0;
            }
        }
        // the last R tokens are still in the window
        for (int q = ((PART && whole) ? len : (len - R > 0 ? len - R : 0)); q < len; ++q) {
            const int32_t v = __builtin_amdgcn_readfirstlane(tok_c(q));
            if (v < 0) continue;
            const unsigned long long hq = __builtin_amdgcn_ballot_w64(slot_node == v);
            if (!hq) continue;                                   // a hot row: never cached
            const int s = (int)__builtin_ctzll(hq);
            const int refc = __builtin_amdgcn_readlane(slot_ref, s) - 1;
            if (lane == s) slot_ref = refc;
            if (refc != 0) continue;
            float l[NV][VEC];
            lds_ld(rowsL + (size_t)s * RW, l);
            if constexpr (RELOAD) {        // what this wavefront changed, added to the row as it is now -- like every row that leaves the window mid-walk (round 6: was load + store)
                float o[NV][VEC];
                o_ld(s, o);
                float *prow = A.SynPos + (int64_t)v * d;
#pragma unroll
                for (int c = 0; c < NV; ++c)
#pragma unroll
                    for (int k = 0; k < VEC; ++k)
                        if ((c * WAVE + lane) * VEC + k < dg)
                            __builtin_amdgcn_global_atomic_fadd_f32((__attribute__((address_space(1))) float *)(prow + (c * WAVE + lane) * VEC + k), l[c][k] - o[c][k]);
            } else {
                if constexpr (DELTA) {
                    float o[NV][VEC], g[NV][VEC];
                    o_ld(s, o);
                    g_ld(A.SynPos + (int64_t)v * d, g);
#pragma unroll
                    for (int c = 0; c < NV; ++c)
#pragma unroll
                        for (int k = 0; k < VEC; ++k) l[c][k] = g[c][k] + (l[c][k] - o[c][k]);
                }
                g_st(A.SynPos + (int64_t)v * d, l);
            }
            if (lane == s) slot_node = -1;
        }
        __builtin_amdgcn_wave_barrier();
    }
    if (lane == 0 && A.pairs) atomicAdd(A.pairs, npairs);
#ifdef GEMHIP_SGNS_PROFILE
    PROF_LAP(0);
    if (lane == 0 && A.prof) for (int k = 0; k < 8; ++k) atomicAdd(A.prof + k, prof_acc[k]);
#endif
}

template <int VEC, int NV>
void launch_sgns(const SgnsArgs &A, int blocks, int threads, size_t lds, hipStream_t s)
{
    hipLaunchKernelGGL((sgns_kernel<VEC, NV>), dim3(blocks), dim3(threads), lds, s, A);
}
template <int VEC, int NV, bool DELTA>
void launch_sgns_win(const SgnsArgs &A, int blocks, int threads, size_t lds, hipStream_t s)
{
    const bool full = A.d == NV * VEC * WAVE, allc = A.cache_radius >= A.window && A.hot_thr == 0;
#define GEMHIP_LAUNCH_WIN(F, C, P, R) hipLaunchKernelGGL((sgns_win_kernel<VEC, NV, DELTA, F, C, P, R>), dim3(blocks), dim3(threads), lds, s, A)
    // instantiations: the A/B knobs (prefetch distance 1, Hogwild WITHOUT reload-on-update) exist for the benchmark shape only (d = 128: VEC 2, NV 1, whole
    // window cached); every other Hogwild launch is reload-on-update with prefetch distance 2
    constexpr bool AB = VEC == 2 && NV == 1;
    const bool reload = DELTA && (A.reload || !(AB && full && allc));
    if constexpr (DELTA) {
        if (reload) {
            if constexpr (AB) { if (full && allc && A.prefetch == 1) { GEMHIP_LAUNCH_WIN(true, true, 1, true); return; } }
            if (full && allc) GEMHIP_LAUNCH_WIN(true, true, 2, true);
            else if (full) GEMHIP_LAUNCH_WIN(true, false, 2, true);
            else if (allc) GEMHIP_LAUNCH_WIN(false, true, 2, true);
            else GEMHIP_LAUNCH_WIN(false, false, 2, true);
            return;
        }
        if constexpr (AB) {      // (only reached with full && allc)
            if (A.prefetch == 1) GEMHIP_LAUNCH_WIN(true, true, 1, false); else GEMHIP_LAUNCH_WIN(true, true, 2, false);
        }
        return;
    } else {
        if constexpr (AB) { if (full && allc && A.prefetch == 1) { GEMHIP_LAUNCH_WIN(true, true, 1, false); return; } }
        if (full && allc) GEMHIP_LAUNCH_WIN(true, true, 2, false);
        else if (full) GEMHIP_LAUNCH_WIN(true, false, 2, false);
        else if (allc) GEMHIP_LAUNCH_WIN(false, true, 2, false);
        else GEMHIP_LAUNCH_WIN(false, false, 2, false);
    }
#undef GEMHIP_LAUNCH_WIN
}
// one bucket of the partitioned schedule: prefetch distance 2; Hogwild launches are reload-on-update (the only Hogwild mode the multi-GPU path uses)
template <int VEC, int NV, bool DELTA>
void launch_sgns_win_part(const SgnsArgs &A, int blocks, int threads, size_t lds, hipStream_t s)
{
    const bool full = A.d == NV * VEC * WAVE, allc = A.cache_radius >= A.window && A.hot_thr == 0;
#define GEMHIP_LAUNCH_PART(F, C) hipLaunchKernelGGL((sgns_win_kernel<VEC, NV, DELTA, F, C, 2, DELTA, true>), dim3(blocks), dim3(threads), lds, s, A)
    if (full && allc) GEMHIP_LAUNCH_PART(true, true);
    else if (full) GEMHIP_LAUNCH_PART(true, false);
    else if (allc) GEMHIP_LAUNCH_PART(false, true);
    else GEMHIP_LAUNCH_PART(false, false);
#undef GEMHIP_LAUNCH_PART
}
}  // namespace
