// gf.hip -- Graph Factorization edge-SGD for MI355X (gfx950).
//
// Replaces gem/embedding/gf.py:93-100 (Python hot loop) and
// gem/c_src/gf.cpp:152-164 (the same loop in the `gf` executable).
//
// Reference semantics (kept exactly):
//   for sweep in range(max_iter):
//     for (i, j, w) in graph.edges():          # grouped by source i, in node-insertion order
//       if j <= i: continue
//       X[i] -= eta * (regu * X[i] - (w - X[i].X[j]) * X[j])     # only row i is written
//
// Device schedule.  Because only the SOURCE row is written and only edges with
// j > i fire, one wavefront can own row i for a whole sweep: X_i lives in
// registers (d=128 -> one float2 per lane), every neighbour row X_j is ONE
// coalesced 4*d-byte read, the dot product is a DPP/readlane wave reduction and
// there are no atomics and no write conflicts.  The sequential (Gauss-Seidel)
// order of the reference is reproduced with two copies of the table:
//   * a row i that the reference visits BEFORE row j reads X_old[j]  (j not yet updated),
//   * a row i visited AFTER row j (j earlier in graph.nodes order but j > i) reads
//     X_new[j]; such rows are placed in a later LEVEL (kernel launch) than j.
// With nodes inserted in ascending id order (every synthetic benchmark graph)
// all rows are level 0 and a sweep is a single launch.  The result differs from
// the fp32 CPU loop only by the summation order inside the dot product.
//
// HBM traffic per sweep (algorithmic, d=128): 512 B read + 512 B write per active
// row, 512 B gather + 8 B (col,w) per update.
#include "common.hpp"
#include <hip/hip_cooperative_groups.h>
#include <cstdlib>
#include <vector>
#include <algorithm>
#include <cmath>

using namespace gemhip;

struct gemhip_gf_plan {
    int64_t n = 0, d = 0, nrows = 0, nupd = 0;
    int device = 0;
    std::vector<int64_t> level_off;   // rows of level L are [level_off[L], level_off[L+1])
    std::vector<int64_t> level_hubs;  // ... of which the first level_hubs[L] are hub rows (gf_hub_kernel)
    std::vector<int64_t> level_maxlen;  // longest non-hub row of the level, in firing edges (the rows-per-wavefront rule looks at it)
    hipStream_t hub_stream = nullptr; hipEvent_t hub_fork = nullptr, hub_join = nullptr;
    int32_t *d_rows = nullptr;        // row ids in processing order (sorted by level, then reference order)
    int64_t *d_ptr = nullptr;         // CSR offsets over d_rows
    uint32_t *d_col = nullptr;        // neighbour id | (1u<<31 if that neighbour is read from X_new)
    float *d_w = nullptr;
    float *X[2] = {nullptr, nullptr};
    bool own_X = false;
    int cur = 0;                      // X[cur] holds the latest table
    int rows_per_wave = 0;            // 0 = auto (gf_rows_per_wave), else forced (gemhip_gf_plan_set_rows_per_wave: tests, A/B)
    // sweeps per cooperative launch (gf_sweeps_coop_kernel): 0 = off (one launch per sweep and level), k > 1 = up to k sweeps per launch on single-level
    // plans without hub rows; fused_grid caps the resident grid (0 = all the occupancy allows).  GEMHIP_GF_FUSED_SWEEPS / _GRID, gemhip_gf_plan_set_fused_sweeps
    int fused_sweeps = getenv("GEMHIP_GF_FUSED_SWEEPS") ? atoi(getenv("GEMHIP_GF_FUSED_SWEEPS")) : 0;
    int fused_grid = getenv("GEMHIP_GF_FUSED_GRID") ? atoi(getenv("GEMHIP_GF_FUSED_GRID")) : 0;
    // Non-temporal hints of the sweep kernels (GEMHIP_GF_NT_STORE, read per plan; -1 = auto).  bit 1 (value 2): the load of a wave's OWN row -- with rows
    // visited in ascending order it is the row's last use of the sweep (only lower rows gather it, and they ran before), and at SBM 1M/10M an XCD's 4 MB
    // of L2 cannot even hold the 5 MB neighbour set the gathers hit in: marking the own-row stream evict-first took a sweep from 548 to 515 us
    // (profiles/r04_ab_gf_nt.jsonl); bit 0 (value 1): the row stores (measured: no effect); bit 2 (value 4): the (col, w) / row-id / offset streams.
    // Auto: 2 on the K-rows-per-wavefront kernel (the big levels), 0 on the one-row kernel (small, L2-resident levels, where nothing needs evicting).
    // A hint only: results are bit-identical either way.
    int nt_store = getenv("GEMHIP_GF_NT_STORE") ? atoi(getenv("GEMHIP_GF_NT_STORE")) : -1;
};

namespace {

constexpr int GF_BLOCK = 256;              // 4 waves, one row per wave
constexpr int GF_WAVES = GF_BLOCK / WAVE;
constexpr int GF_PREFETCH = 4;             // neighbour rows in flight per wave
constexpr int GF_HUB_EDGES = 1024;         // rows with at least this many firing edges get a workgroup (gf_hub_kernel)
constexpr int GF_PREFETCH_DEEP = 16;       // ... on full 64-edge chunks of long (hub) rows, divided by the registers a row needs

template <int VEC> struct vec_t;
template <> struct vec_t<1> { using type = float; };
template <> struct vec_t<2> { using type = float2; };

template <int VEC>
__device__ __forceinline__ void load_row(const float *__restrict__ p, int d, int lane, int c, float (&v)[VEC])
{
    const int idx = (c * WAVE + lane) * VEC;
    if constexpr (VEC == 2) {
        if (idx < d) { const float2 t = *reinterpret_cast<const float2 *>(p + idx); v[0] = t.x; v[1] = t.y; }
        else { v[0] = 0.f; v[1] = 0.f; }
    } else {
        v[0] = idx < d ? p[idx] : 0.f;
    }
}

// a wave's OWN row: with rows visited in ascending order its last use of the sweep (only lower rows gather it, and they ran before) -- bit 2 of the
// plan's nt_store asks for the non-temporal hint on this load as well
template <int VEC>
__device__ __forceinline__ void load_own_row(const float *__restrict__ p, int d, int lane, int c, float (&v)[VEC], int nt)
{
    const int idx = (c * WAVE + lane) * VEC;
    if (!(nt & 2)) { load_row<VEC>(p, d, lane, c, v); return; }
    if constexpr (VEC == 2) {
        typedef float gf_f2 __attribute__((ext_vector_type(2)));
        if (idx < d) { const gf_f2 t = __builtin_nontemporal_load(reinterpret_cast<const gf_f2 *>(p + idx)); v[0] = t.x; v[1] = t.y; }
        else { v[0] = 0.f; v[1] = 0.f; }
    } else {
        v[0] = idx < d ? __builtin_nontemporal_load(p + idx) : 0.f;
    }
}

template <int VEC, int NV>
__device__ __forceinline__ void store_row(float *po, int d, int lane, const float (&xi)[NV][VEC], int nt)
{
#pragma unroll
    for (int c = 0; c < NV; ++c) {
        const int idx = (c * WAVE + lane) * VEC;
        if (idx < d) {
            if constexpr (VEC == 2) {
                typedef float gf_f2 __attribute__((ext_vector_type(2)));
                gf_f2 v; v.x = xi[c][0]; v.y = xi[c][1];
                if (nt & 1) __builtin_nontemporal_store(v, reinterpret_cast<gf_f2 *>(po + idx));
                else *reinterpret_cast<gf_f2 *>(po + idx) = v;
            } else {
                if (nt & 1) __builtin_nontemporal_store(xi[c][0], po + idx);
                else po[idx] = xi[c][0];
            }
        }
    }
}

// One edge update X_i -= eta * (regu * X_i - (w_ij - X_i.X_j) * X_j)  (gf.cpp:162-163).  Shared by the wave-per-row kernel and the hub
// kernel: the same expressions are contracted the same way, so both produce bit-identical rows.
template <int VEC, int NV>
__device__ __forceinline__ void gf_apply_edge(float (&xi)[NV][VEC], const float (&xj)[NV][VEC], float wij, float eta, float regu)
{
    // explicit fused / rounded operations: the compiler may not choose between mul+add and fma (or packed forms) differently per kernel
    float part = 0.f;
#pragma unroll
    for (int q = 0; q < NV; ++q)
#pragma unroll
        for (int v = 0; v < VEC; ++v) part = fmaf(xi[q][v], xj[q][v], part);
    const float coef = wij - wave_sum(part);        // (w_ij - X_i.X_j)
#pragma unroll
    for (int q = 0; q < NV; ++q)
#pragma unroll
        for (int v = 0; v < VEC; ++v) {
            const float t = fmaf(-coef, xj[q][v], __fmul_rn(regu, xi[q][v]));     // regu * X_i - coef * X_j
            xi[q][v] = fmaf(-eta, t, xi[q][v]);
        }
}

// Apply the updates of up to 64 edges of one row (columns/weights in lane registers cj/wj), U neighbour rows in flight.
// (A rolling window that re-issues a load right after each consumed row measured 6-15 % SLOWER than these plain batches; U = 2 / 4 / 6 / 8 at SBM 1M/10M:
// 0.648 / 0.598 / 0.598 / 0.610 ms per sweep.)
template <int VEC, int NV, int U>
__device__ __forceinline__ void gf_chunk(float (&xi)[NV][VEC], uint32_t cj, float wj, int cnt, const float *Xold, const float *Xnew, int d, int lane,
                                         float eta, float regu)
{
    for (int k = 0; k < cnt; k += U) {
        float xj[U][NV][VEC];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int kk = (k + u) < cnt ? (k + u) : (cnt - 1);
            const uint32_t c = bcast_lane(cj, kk);
            const float *pj = ((c >> 31) ? Xnew : Xold) + (int64_t)(c & 0x7fffffffu) * d;
#pragma unroll
            for (int q = 0; q < NV; ++q) load_row<VEC>(pj, d, lane, q, xj[u][q]);
        }
#pragma unroll
        for (int u = 0; u < U; ++u)
            if (k + u < cnt) gf_apply_edge<VEC, NV>(xi, xj[u], bcast_lane(wj, k + u), eta, regu);
    }
}

// One sweep over the rows of one level.  One wavefront per row.
template <int VEC, int NV>
__global__ __launch_bounds__(GF_BLOCK) void gf_sweep_kernel(const int32_t *__restrict__ rows, const int64_t *__restrict__ ptr,
                                                            const uint32_t *__restrict__ col, const float *__restrict__ w,
                                                            const float *Xold, float *Xnew, int64_t row0, int64_t nrows, int d,
                                                            float eta, float regu, int nt)
{
    const int lane = lane_id();
    const int wave = threadIdx.x >> 6;
    const int64_t slot = xcd_contiguous_block(blockIdx.x, gridDim.x) * GF_WAVES + wave;
    if (slot >= nrows) return;
    const int64_t r = row0 + slot;
    const int32_t i = rows[r];
    const int64_t e0 = ptr[r], e1 = ptr[r + 1];

    float xi[NV][VEC];
    const float *pi = Xold + (int64_t)i * d;
#pragma unroll
    for (int c = 0; c < NV; ++c) load_own_row<VEC>(pi, d, lane, c, xi[c], nt);

    // hub rows (power-law graphs) are one long dependent chain of updates: only memory latency can be hidden, so full
    // 64-edge chunks keep GF_PREFETCH_DEEP neighbour rows in flight instead of GF_PREFETCH
    constexpr int DEEP = (GF_PREFETCH_DEEP / NV) >= GF_PREFETCH ? (GF_PREFETCH_DEEP / NV) : GF_PREFETCH;
    for (int64_t e = e0; e < e1; e += WAVE) {
        const int cnt = (int)((e1 - e) < (int64_t)WAVE ? (e1 - e) : (int64_t)WAVE);
        // coalesced read of up to 64 (col, w) pairs of this row; broadcast later with v_readlane
        const uint32_t cj = lane < cnt ? col[e + lane] : 0u;
        const float wj = lane < cnt ? w[e + lane] : 0.f;
        if (cnt == WAVE) gf_chunk<VEC, NV, DEEP>(xi, cj, wj, cnt, Xold, Xnew, d, lane, eta, regu);
        else gf_chunk<VEC, NV, GF_PREFETCH>(xi, cj, wj, cnt, Xold, Xnew, d, lane, eta, regu);
    }
    store_row<VEC, NV>(Xnew + (int64_t)i * d, d, lane, xi, nt);
}

// Several sweeps in ONE launch (round 5; experiment behind gemhip_gf_plan_set_fused_sweeps / GEMHIP_GF_FUSED_SWEEPS).  At SBM 10k/100k (BASELINE configs[1])
// a sweep is a ~5 us kernel and 1000 of them cost 9 us each: the loop is launch-bound.  This kernel keeps the grid resident (cooperative launch: every
// workgroup co-resident by construction) and separates the sweeps with cooperative_groups' grid barrier, which carries the agent-scope release / acquire
// that makes the rows one XCD wrote visible to the other seven (L2 write-back + invalidate: what a kernel boundary does).  Same row body as
// gf_sweep_kernel, same edge order: bit-identical tables.  Single-level plans without hub rows only (every benchmark graph).
template <int VEC, int NV>
__global__ __launch_bounds__(GF_BLOCK) void gf_sweeps_coop_kernel(const int32_t *__restrict__ rows, const int64_t *__restrict__ ptr,
                                                                  const uint32_t *__restrict__ col, const float *__restrict__ w, float *X0, float *X1,
                                                                  int64_t nrows, int d, float eta, float regu, int nsweeps)
{
    cooperative_groups::grid_group grid = cooperative_groups::this_grid();
    const int lane = lane_id();
    const int wave = threadIdx.x >> 6;
    const int64_t nslots = (int64_t)gridDim.x * GF_WAVES;
    const int64_t first = xcd_contiguous_block(blockIdx.x, gridDim.x) * GF_WAVES + wave;
    constexpr int DEEP = (GF_PREFETCH_DEEP / NV) >= GF_PREFETCH ? (GF_PREFETCH_DEEP / NV) : GF_PREFETCH;
    for (int s = 0; s < nsweeps; ++s) {
        const float *Xold = (s & 1) ? X1 : X0;
        float *Xnew = (s & 1) ? X0 : X1;
        for (int64_t r = first; r < nrows; r += nslots) {
            const int32_t i = rows[r];
            const int64_t e0 = ptr[r], e1 = ptr[r + 1];
            float xi[NV][VEC];
            const float *pi = Xold + (int64_t)i * d;
#pragma unroll
            for (int c = 0; c < NV; ++c) load_row<VEC>(pi, d, lane, c, xi[c]);
            for (int64_t e = e0; e < e1; e += WAVE) {
                const int cnt = (int)((e1 - e) < (int64_t)WAVE ? (e1 - e) : (int64_t)WAVE);
                const uint32_t cj = lane < cnt ? col[e + lane] : 0u;
                const float wj = lane < cnt ? w[e + lane] : 0.f;
                if (cnt == WAVE) gf_chunk<VEC, NV, DEEP>(xi, cj, wj, cnt, Xold, Xnew, d, lane, eta, regu);
                else gf_chunk<VEC, NV, GF_PREFETCH>(xi, cj, wj, cnt, Xold, Xnew, d, lane, eta, regu);
            }
            store_row<VEC, NV>(Xnew + (int64_t)i * d, d, lane, xi, 0);
        }
        grid.sync();
    }
}

template <int VEC, int NV>
int launch_coop(const gemhip_gf_plan *p, float *X0, float *X1, float eta, float regu, int nsweeps, hipStream_t s)
{
    // resident workgroups of this instantiation on this device: queried once (both calls are millisecond-scale host work, and the fused path launches
    // once per `fused_sweeps` sweeps -- ADVICE r5).  The kernel reads Xold only: the caller takes this path for single-level plans alone (nlevels == 1 at
    // the launch site), and those never set the "read Xnew" column bit (bit 31).
    static int64_t resident = 0;
    if (resident == 0) {
        int per_cu = 0, dev = 0;
        hipDeviceProp_t prop;
        GEMHIP_CHECK(hipGetDevice(&dev));
        GEMHIP_CHECK(hipGetDeviceProperties(&prop, dev));
        GEMHIP_CHECK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, gf_sweeps_coop_kernel<VEC, NV>, GF_BLOCK, 0));
        resident = (int64_t)per_cu * prop.multiProcessorCount;
    }
    int64_t grid = std::min<int64_t>(resident, (p->nrows + GF_WAVES - 1) / GF_WAVES);
    grid = std::max<int64_t>(NUM_XCD, grid / NUM_XCD * NUM_XCD);              // a multiple of 8: the XCD-contiguous map covers every slot
    if (p->fused_grid > 0) grid = std::max<int64_t>(NUM_XCD, std::min<int64_t>(grid, (int64_t)p->fused_grid / NUM_XCD * NUM_XCD));
    const int32_t *rows = p->d_rows; const int64_t *ptr = p->d_ptr; const uint32_t *col = p->d_col; const float *w = p->d_w;
    int64_t nrows = p->nrows; int d = (int)p->d;
    void *args[] = {(void *)&rows, (void *)&ptr, (void *)&col, (void *)&w, (void *)&X0, (void *)&X1, (void *)&nrows, (void *)&d, (void *)&eta, (void *)&regu, (void *)&nsweeps};
    GEMHIP_CHECK(hipLaunchCooperativeKernel((const void *)gf_sweeps_coop_kernel<VEC, NV>, dim3((unsigned)grid), dim3(GF_BLOCK), args, 0, s));
    return GEMHIP_OK;
}

// The same sweep with K consecutive rows per wavefront (large levels; round 4).  A wave that owns ONE row spends its life in a chain of dependent
// round trips -- row id and edge offsets, then (col, w) and X_i, then the neighbour rows, then the store -- and only the third moves data: at SBM
// 1M/10M the kernel reached 3.3 TB/s of real traffic (0.53 of a copy).  Here lane k of a wave fetches the id and the offsets of its k-th row in ONE
// load, and while row k is trained the (col, w) chunk and X_i of row k+1 are already in flight, so the only exposed latency per row is the gather of
// its neighbour rows.  Same gf_apply_edge, same edge order inside a row: bit-identical to gf_sweep_kernel (rows of one level are independent).
template <int VEC, int NV>
__global__ __launch_bounds__(GF_BLOCK) void gf_sweep_rows_kernel(const int32_t *__restrict__ rows, const int64_t *__restrict__ ptr,
                                                                 const uint32_t *__restrict__ col, const float *__restrict__ w,
                                                                 const float *Xold, float *Xnew, int64_t row0, int64_t nrows, int d,
                                                                 float eta, float regu, int K, int nt)
{
    const int lane = lane_id();
    const int wave = threadIdx.x >> 6;
    // K CONSECUTIVE rows.  (Measured and dropped, round 4: striping a wavefront's rows 512 / 1024 / 2048 apart inside super-blocks, so that the
    // wavefronts resident on an XCD work in one band of rows at a time -- 560 / 567 / 567 us against 564: the L2 misses of the neighbour gathers are
    // capacity misses of a 5 MB block in 4 MB of L2, not a matter of which rows run together.)
    const int64_t first = (xcd_contiguous_block(blockIdx.x, gridDim.x) * GF_WAVES + wave) * K;
    if (first >= nrows) return;
    const int nk = (int)((nrows - first) < (int64_t)K ? (nrows - first) : (int64_t)K);
    int32_t rv = 0; int64_t pa = 0, pb = 0;
    if (lane < nk) {
        if (nt & 4) { rv = __builtin_nontemporal_load(rows + row0 + first + lane); pa = __builtin_nontemporal_load(ptr + row0 + first + lane); pb = __builtin_nontemporal_load(ptr + row0 + first + lane + 1); }
        else { rv = rows[row0 + first + lane]; pa = ptr[row0 + first + lane]; pb = ptr[row0 + first + lane + 1]; }
    }
    auto ld_col = [&](int64_t e) -> uint32_t { return (nt & 4) ? __builtin_nontemporal_load(col + e) : col[e]; };
    auto ld_w = [&](int64_t e) -> float { return (nt & 4) ? __builtin_nontemporal_load(w + e) : w[e]; };
    auto lane64 = [&](int64_t v, int k) -> int64_t {
        const uint32_t lo = bcast_lane((uint32_t)v, k), hi = bcast_lane((uint32_t)((uint64_t)v >> 32), k);
        return (int64_t)(((uint64_t)hi << 32) | lo);
    };
    constexpr int DEEP = (GF_PREFETCH_DEEP / NV) >= GF_PREFETCH ? (GF_PREFETCH_DEEP / NV) : GF_PREFETCH;
    // row 0: first (col, w) chunk and X_i
    int32_t i_n = bcast_lane(rv, 0);
    int64_t e0_n = lane64(pa, 0), e1_n = lane64(pb, 0);
    int cnt_n = (int)((e1_n - e0_n) < (int64_t)WAVE ? (e1_n - e0_n) : (int64_t)WAVE);
    uint32_t cj_n = lane < cnt_n ? ld_col(e0_n + lane) : 0u;
    float wj_n = lane < cnt_n ? ld_w(e0_n + lane) : 0.f;
    float xi_n[NV][VEC];
#pragma unroll
    for (int c = 0; c < NV; ++c) load_own_row<VEC>(Xold + (int64_t)i_n * d, d, lane, c, xi_n[c], nt);
    for (int k = 0; k < nk; ++k) {
        const int32_t i = i_n;
        const int64_t e0 = e0_n, e1 = e1_n;
        const int cnt0 = cnt_n;
        const uint32_t cj0 = cj_n;
        const float wj0 = wj_n;
        float xi[NV][VEC];
#pragma unroll
        for (int c = 0; c < NV; ++c)
#pragma unroll
            for (int v = 0; v < VEC; ++v) xi[c][v] = xi_n[c][v];
        if (k + 1 < nk) {                                // the next row's inputs travel while this row is trained
            i_n = bcast_lane(rv, k + 1);
            e0_n = lane64(pa, k + 1); e1_n = lane64(pb, k + 1);
            cnt_n = (int)((e1_n - e0_n) < (int64_t)WAVE ? (e1_n - e0_n) : (int64_t)WAVE);
            cj_n = lane < cnt_n ? ld_col(e0_n + lane) : 0u;
            wj_n = lane < cnt_n ? ld_w(e0_n + lane) : 0.f;
#pragma unroll
            for (int c = 0; c < NV; ++c) load_own_row<VEC>(Xold + (int64_t)i_n * d, d, lane, c, xi_n[c], nt);
        }
        if (cnt0 == WAVE) gf_chunk<VEC, NV, DEEP>(xi, cj0, wj0, cnt0, Xold, Xnew, d, lane, eta, regu);
        else if (cnt0 > 0) gf_chunk<VEC, NV, GF_PREFETCH>(xi, cj0, wj0, cnt0, Xold, Xnew, d, lane, eta, regu);
        for (int64_t e = e0 + WAVE; e < e1; e += WAVE) {
            const int cnt = (int)((e1 - e) < (int64_t)WAVE ? (e1 - e) : (int64_t)WAVE);
            const uint32_t cj = lane < cnt ? ld_col(e + lane) : 0u;
            const float wj = lane < cnt ? ld_w(e + lane) : 0.f;
            if (cnt == WAVE) gf_chunk<VEC, NV, DEEP>(xi, cj, wj, cnt, Xold, Xnew, d, lane, eta, regu);
            else gf_chunk<VEC, NV, GF_PREFETCH>(xi, cj, wj, cnt, Xold, Xnew, d, lane, eta, regu);
        }
        store_row<VEC, NV>(Xnew + (int64_t)i * d, d, lane, xi, nt);
    }
}

// rows per wavefront of a level with `nrows` rows: 1 (gf_sweep_kernel) until every resident wave slot of the chip (256 CUs x 32 waves) has two rows
// to work on, then up to GEMHIP_GF_ROWS_PER_WAVE (default 8; read once): a level of 946 188 rows (SBM 1M/10M) runs 8 rows per wave
int gf_rows_per_wave(int64_t nrows)
{
    static const int kmax = getenv("GEMHIP_GF_ROWS_PER_WAVE") ? std::max(1, std::min(64, atoi(getenv("GEMHIP_GF_ROWS_PER_WAVE")))) : 8;
    const int64_t k = nrows / (2 * 256 * 32);
    return (int)std::max<int64_t>(1, std::min<int64_t>(k, kmax));
}

template <int VEC, int NV>
void launch_sweep(const gemhip_gf_plan *p, int64_t row0, int64_t nrows, const float *Xold, float *Xnew, float eta, float regu,
                  hipStream_t s)
{
    // K rows per wavefront pays where rows are short and alike (SBM: 548 against 579 us per sweep at 1M/10M); on a power-law level a wavefront that
    // draws a few long rows among its K holds the launch up (R-MAT scale 22: 6.99 against 6.52 ms) -- levels with rows of more than two 64-edge
    // chunks keep one row per wavefront
    int64_t maxlen = 0;
    for (size_t l = 0; l + 1 < p->level_off.size(); ++l)
        if (row0 >= p->level_off[l] && row0 < p->level_off[l + 1]) maxlen = p->level_maxlen.size() > l ? p->level_maxlen[l] : 0;
    const int K = p->rows_per_wave > 0 ? p->rows_per_wave : (maxlen > 2 * WAVE ? 1 : gf_rows_per_wave(nrows));
    if (K > 1) {
        const int64_t waves = (nrows + K - 1) / K;
        const int64_t blocks = (waves + GF_WAVES - 1) / GF_WAVES;
        const int64_t grid = (blocks + NUM_XCD - 1) / NUM_XCD * NUM_XCD;
        hipLaunchKernelGGL((gf_sweep_rows_kernel<VEC, NV>), dim3((unsigned)grid), dim3(GF_BLOCK), 0, s, p->d_rows, p->d_ptr, p->d_col,
                           p->d_w, Xold, Xnew, row0, nrows, (int)p->d, eta, regu, K, p->nt_store < 0 ? 2 : p->nt_store);
        return;
    }
    const int64_t blocks = (nrows + GF_WAVES - 1) / GF_WAVES;
    // round the grid up to a multiple of 8 so the XCD-contiguous map covers every slot
    const int64_t grid = (blocks + NUM_XCD - 1) / NUM_XCD * NUM_XCD;
    hipLaunchKernelGGL((gf_sweep_kernel<VEC, NV>), dim3((unsigned)grid), dim3(GF_BLOCK), 0, s, p->d_rows, p->d_ptr, p->d_col,
                       p->d_w, Xold, Xnew, row0, nrows, (int)p->d, eta, regu, p->nt_store < 0 ? 0 : p->nt_store);
}

// Hub rows (power-law graphs).  The updates of one row are one dependent chain (exact Gauss-Seidel): a wave that also fetches its
// neighbour rows pays a memory latency per batch on top of the chain (130 ns per update measured).  Here a row with >= GF_HUB_EDGES firing
// edges gets a whole workgroup: wavefronts 1-NP stream the neighbour rows (and weights) of successive batches through an LDS ring, wave 0
// does nothing but apply them, in edge order, with the very same gf_apply_edge -- bit-identical to the wave-per-row kernel.
typedef __attribute__((address_space(3))) volatile int32_t gf_lds_vi32;
__device__ __forceinline__ int32_t gf_flag_ld(const int32_t *p) { return *(gf_lds_vi32 *)p; }
__device__ __forceinline__ void gf_flag_st(int32_t *p, int32_t v) { *(gf_lds_vi32 *)p = v; }

template <int VEC, int NV, int NP>
__global__ __launch_bounds__((NP + 1) * 64) void gf_hub_kernel(const int32_t *__restrict__ rows, const int64_t *__restrict__ ptr,
                                                     const uint32_t *__restrict__ col, const float *__restrict__ w, const float *Xold,
                                                     float *Xnew, int64_t row0, int d, float eta, float regu)
{
    constexpr int RW = NV * VEC * WAVE;              // floats of one staged row
    constexpr int B = NV >= 8 ? 2 : 16 / NV;         // edges per batch
    constexpr int NBATCH = NP + 1;                   // ring depth in batches: one per producer wavefront + the one being consumed (8 KB of rows each)
    constexpr int G = B >= 4 ? 4 : B;                // rows the consumer moves LDS -> registers at a time
    __shared__ __attribute__((aligned(16))) float ring[NBATCH * B * RW];
    __shared__ float wring[NBATCH * 16];
    __shared__ int32_t ready[NBATCH];
    __shared__ int32_t done;
    const int lane = lane_id();
    const int wave = threadIdx.x >> 6;
    const int64_t r = row0 + blockIdx.x;
    const int32_t i = rows[r];
    const int64_t e0 = ptr[r], e1 = ptr[r + 1];
    const int nb = (int)((e1 - e0 + B - 1) / B);
    if (threadIdx.x == 0) done = 0;
    if (threadIdx.x < NBATCH) ready[threadIdx.x] = 0;
    __syncthreads();
    if (wave == 0) {
        float xi[NV][VEC];
        const float *pi = Xold + (int64_t)i * d;
#pragma unroll
        for (int c = 0; c < NV; ++c) load_row<VEC>(pi, d, lane, c, xi[c]);
        auto lds_row = [&](const float *row, float (&v)[NV][VEC]) {
#pragma unroll
            for (int q = 0; q < NV; ++q)
#pragma unroll
                for (int k = 0; k < VEC; ++k) v[q][k] = row[(q * WAVE + lane) * VEC + k];
        };
        for (int t = 0; t < nb; ++t) {
            const int sb = t % NBATCH;
            while (gf_flag_ld(&ready[sb]) != t + 1) __builtin_amdgcn_s_sleep(0);
            asm volatile("" ::: "memory");
            const int cnt = (int)((e1 - (e0 + (int64_t)t * B)) < B ? (e1 - (e0 + (int64_t)t * B)) : B);
            const float wv = wring[sb * 16 + (lane & 15)];
            const float *base = ring + (size_t)sb * B * RW;
            float buf[2][G][NV][VEC];
#pragma unroll
            for (int u = 0; u < G; ++u) lds_row(base + (size_t)u * RW, buf[0][u]);
#pragma unroll
            for (int g = 0; g < B / G; ++g) {
                if (g + 1 < B / G) {
#pragma unroll
                    for (int u = 0; u < G; ++u) lds_row(base + (size_t)((g + 1) * G + u) * RW, buf[(g + 1) & 1][u]);
                }
#pragma unroll
                for (int u = 0; u < G; ++u)
                    if (g * G + u < cnt) gf_apply_edge<VEC, NV>(xi, buf[g & 1][u], bcast_lane(wv, g * G + u), eta, regu);
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            if (lane == 0) gf_flag_st(&done, t + 1);
        }
        float *po = Xnew + (int64_t)i * d;
#pragma unroll
        for (int c = 0; c < NV; ++c) {
            const int idx = (c * WAVE + lane) * VEC;
            if (idx < d) {
                if constexpr (VEC == 2) *reinterpret_cast<float2 *>(po + idx) = make_float2(xi[c][0], xi[c][1]);
                else po[idx] = xi[c][0];
            }
        }
    } else {
        for (int t = wave - 1; t < nb; t += NP) {
            const int sb = t % NBATCH;
            while (gf_flag_ld(&done) < t - NBATCH + 1) __builtin_amdgcn_s_sleep(1);     // the batch that used this ring segment is consumed
            asm volatile("" ::: "memory");
            const int64_t eb = e0 + (int64_t)t * B;
            const int cnt = (int)((e1 - eb) < B ? (e1 - eb) : B);
            const uint32_t cj = lane < cnt ? col[eb + lane] : 0u;
            const float wj = lane < cnt ? w[eb + lane] : 0.f;
            float xj[B][NV][VEC];
#pragma unroll
            for (int u = 0; u < B; ++u) {
                const int kk = u < cnt ? u : cnt - 1;
                const uint32_t c = bcast_lane(cj, kk);
                const float *pj = ((c >> 31) ? Xnew : Xold) + (int64_t)(c & 0x7fffffffu) * d;
#pragma unroll
                for (int q = 0; q < NV; ++q) load_row<VEC>(pj, d, lane, q, xj[u][q]);
            }
            float *base = ring + (size_t)sb * B * RW;
#pragma unroll
            for (int u = 0; u < B; ++u)
#pragma unroll
                for (int q = 0; q < NV; ++q)
#pragma unroll
                    for (int k = 0; k < VEC; ++k) base[(size_t)u * RW + (q * WAVE + lane) * VEC + k] = xj[u][q][k];
            if (lane < 16) wring[sb * 16 + lane] = wj;
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            if (lane == 0) gf_flag_st(&ready[sb], t + 1);
        }
    }
}

template <int VEC, int NV>
void launch_hub(const gemhip_gf_plan *p, int64_t row0, int64_t nhub, const float *Xold, float *Xnew, float eta, float regu, hipStream_t s)
{
    // producer wavefronts per hub row: the chain's wavefront consumes a neighbour row every ~50 ns, a producer needs two loaded-HBM round
    // trips per batch of 16 while the wave-per-row kernel saturates the memory system next to it (GEMHIP_GF_HUB_PRODUCERS=3|7 for the A/B)
    static const int np = getenv("GEMHIP_GF_HUB_PRODUCERS") ? atoi(getenv("GEMHIP_GF_HUB_PRODUCERS")) : 7;
    if (np >= 7)
        hipLaunchKernelGGL((gf_hub_kernel<VEC, NV, 7>), dim3((unsigned)nhub), dim3(512), 0, s, p->d_rows, p->d_ptr, p->d_col, p->d_w, Xold, Xnew, row0,
                           (int)p->d, eta, regu);
    else
        hipLaunchKernelGGL((gf_hub_kernel<VEC, NV, 3>), dim3((unsigned)nhub), dim3(256), 0, s, p->d_rows, p->d_ptr, p->d_col, p->d_w, Xold, Xnew, row0,
                           (int)p->d, eta, regu);
}
using hub_fn = void (*)(const gemhip_gf_plan *, int64_t, int64_t, const float *, float *, float, float, hipStream_t);
hub_fn pick_hub(int d)
{
    if (d % 2 == 0) {
        const int nv = (d + 127) / 128;
        return nv <= 1 ? launch_hub<2, 1> : nv <= 2 ? launch_hub<2, 2> : nv <= 4 ? launch_hub<2, 4> : nv <= 8 ? launch_hub<2, 8> : nullptr;
    }
    const int nv = (d + 63) / 64;
    return nv <= 1 ? launch_hub<1, 1> : nv <= 2 ? launch_hub<1, 2> : nv <= 4 ? launch_hub<1, 4> : nv <= 8 ? launch_hub<1, 8> : nullptr;
}

using sweep_fn = void (*)(const gemhip_gf_plan *, int64_t, int64_t, const float *, float *, float, float, hipStream_t);
using coop_fn = int (*)(const gemhip_gf_plan *, float *, float *, float, float, int, hipStream_t);
coop_fn pick_coop(int d)
{
    if (d % 2 == 0) {
        const int nv = (d + 127) / 128;
        return nv <= 1 ? launch_coop<2, 1> : nv <= 2 ? launch_coop<2, 2> : nv <= 4 ? launch_coop<2, 4> : nv <= 8 ? launch_coop<2, 8> : nullptr;
    }
    const int nv = (d + 63) / 64;
    return nv <= 1 ? launch_coop<1, 1> : nv <= 2 ? launch_coop<1, 2> : nv <= 4 ? launch_coop<1, 4> : nv <= 8 ? launch_coop<1, 8> : nullptr;
}

sweep_fn pick_sweep(int d)
{
    if (d % 2 == 0) {
        const int nv = (d + 127) / 128;
        if (nv <= 1) return launch_sweep<2, 1>;
        if (nv <= 2) return launch_sweep<2, 2>;
        if (nv <= 4) return launch_sweep<2, 4>;
        if (nv <= 8) return launch_sweep<2, 8>;
        return nullptr;
    }
    const int nv = (d + 63) / 64;
    if (nv <= 1) return launch_sweep<1, 1>;
    if (nv <= 2) return launch_sweep<1, 2>;
    if (nv <= 4) return launch_sweep<1, 4>;
    if (nv <= 8) return launch_sweep<1, 8>;
    return nullptr;
}

// 0.01*N(0,1)-style init: thread t fills elements 4t..4t+3 from one Philox block.
__global__ void gf_init_kernel(float *X, int64_t total, uint64_t seed, float scale)
{
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t base = t * 4;
    if (base >= total) return;
    const u32x4 r = philox4x32_10(seed, (uint32_t)t, (uint32_t)(t >> 32), 0x6766u /* 'gf' */, 0u);
    const float u1 = 1.0f - u01(r.x), u2 = u01(r.y), u3 = 1.0f - u01(r.z), u4 = u01(r.w);   // (0,1]
    const float ra = sqrtf(-2.0f * logf(u1)), rb = sqrtf(-2.0f * logf(u3));
    float s0, c0, s1, c1;
    sincosf(6.28318530717958647692f * u2, &s0, &c0);
    sincosf(6.28318530717958647692f * u4, &s1, &c1);
    const float z[4] = {ra * c0, ra * s0, rb * c1, rb * s1};
    for (int k = 0; k < 4; ++k)
        if (base + k < total) X[base + k] = scale * z[k];
}

// gf.cpp:94-113 -- f1 = sum_e (w - X_i.X_j)^2 over ALL edges, f2 = ||X||_F^2.  One wave per edge.
__global__ __launch_bounds__(256) void gf_objective_kernel(const int32_t *__restrict__ src, const int32_t *__restrict__ dst,
                                                           const float *__restrict__ w, const float *__restrict__ X, int64_t m,
                                                           int64_t n, int d, double *out)
{
    const int lane = lane_id();
    const int64_t wave0 = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
    double f1 = 0.0, f2 = 0.0;
    for (int64_t e = wave0; e < m; e += nwaves) {
        const float *xi = X + (int64_t)src[e] * d, *xj = X + (int64_t)dst[e] * d;
        float part = 0.f;
        for (int k = lane; k < d; k += WAVE) part += xi[k] * xj[k];
        const float r = (w ? w[e] : 1.0f) - wave_sum(part);
        f1 += (double)r * (double)r;
    }
    for (int64_t v = wave0; v < n; v += nwaves) {
        const float *x = X + v * d;
        float part = 0.f;
        for (int k = lane; k < d; k += WAVE) part += x[k] * x[k];
        f2 += (double)wave_sum(part);
    }
    if (lane == 0) {
        atomicAdd(out, f1);
        atomicAdd(out + 1, f2);
    }
}

}  // namespace

// ------------------------------------------------------------------- host API
extern "C" int gemhip_gf_plan_create(int64_t n, int64_t m, const int32_t *src, const int32_t *dst, const float *w, int32_t d,
                                     int64_t row_begin, int64_t row_end, gemhip_gf_plan_t *out)
{
    GEMHIP_REQUIRE(out != nullptr, "gf_plan_create: out is NULL");
    *out = nullptr;
    GEMHIP_REQUIRE(n > 0 && n < (int64_t)0x7fffffff, "gf_plan_create: n=%lld out of range", (long long)n);
    GEMHIP_REQUIRE(m >= 0 && (m == 0 || (src && dst)), "gf_plan_create: bad edge arrays");
    GEMHIP_REQUIRE(d >= 1, "gf_plan_create: d=%d", d);
    GEMHIP_REQUIRE(pick_sweep(d) != nullptr, "gf_plan_create: d=%d unsupported (even d <= 1024, odd d <= 512)", d);
    GEMHIP_REQUIRE(0 <= row_begin && row_begin <= row_end && row_end <= n, "gf_plan_create: bad row range");

    // 1. rows in the order the reference first visits them; keep only firing edges (dst > src) of owned rows.
    const double t_host0 = phase_now();
    std::vector<int32_t> pos(n, -1);          // pos[i] = rank of row i among firing source rows (reference order)
    std::vector<int32_t> order;               // row ids by pos
    std::vector<int64_t> deg;
    // The plan gives a source row ONE wavefront and all of its firing edges in one go, reading X_new[j] for neighbours whose
    // row comes earlier in first-visit order and X_old[j] otherwise.  That equals the reference's strictly sequential loop
    // (gf.py:93-100, gf.cpp:152-164) iff every firing edge (i,j) at position t sees either ALL of j's updates of this sweep
    // (j's last firing edge is before t, and j was first visited before i) or NONE (j's first firing edge is after t) --
    // always true when a source's edges are contiguous (graph.edges(), saveGraphToEdgeListTxt), and for many interleaved
    // lists too.  A list that needs an intermediate version of a row (e.g. (1,2),(0,1),(1,3): row 0 must see X_1 between its
    // two updates) has no schedule with two table versions; it is rejected instead of silently reordered.
    std::vector<int64_t> first_t(n, -1), last_t(n, -1);
    for (int64_t e = 0; e < m; ++e) {
        const int32_t i = src[e], j = dst[e];
        GEMHIP_REQUIRE(i >= 0 && i < n && j >= 0 && j < n, "gf_plan_create: edge %lld = (%d,%d) outside [0,%lld)", (long long)e, i, j,
                       (long long)n);
        if (j <= i) continue;                        // does not fire (gf.py:95, gf.cpp:157)
        if (first_t[i] < 0) first_t[i] = e;
        last_t[i] = e;
    }
    for (int64_t e = 0; e < m; ++e) {
        const int32_t i = src[e], j = dst[e];
        if (j <= i || first_t[j] < 0) continue;      // j never fires: its row is the same in both tables
        const bool ok = first_t[j] < first_t[i] ? last_t[j] < e : first_t[j] > e;
        GEMHIP_REQUIRE(ok, "gf_plan_create: edge %lld = (%d,%d) reads row %d while that row is only partly updated in file order "
                       "(its firing edges span positions %lld..%lld); the sequential semantics of gf.cpp need the edges of a source "
                       "to be contiguous (graph.edges() order) -- group the list by source first if the reordering is acceptable",
                       (long long)e, i, j, j, (long long)first_t[j], (long long)last_t[j]);
    }
    for (int64_t e = 0; e < m; ++e) {
        const int32_t i = src[e], j = dst[e];
        if (j <= i || i < row_begin || i >= row_end) continue;
        if (pos[i] < 0) { pos[i] = (int32_t)order.size(); order.push_back(i); deg.push_back(0); }
        ++deg[pos[i]];
    }
    const int64_t nrows = (int64_t)order.size();
    std::vector<int64_t> off(nrows + 1, 0);
    for (int64_t r = 0; r < nrows; ++r) off[r + 1] = off[r] + deg[r];
    const int64_t nupd = off[nrows];
    std::vector<uint32_t> col(nupd);
    std::vector<float> wt(nupd);
    {
        std::vector<int64_t> fill(off.begin(), off.end() - 1);
        for (int64_t e = 0; e < m; ++e) {
            const int32_t i = src[e], j = dst[e];
            if (j <= i || i < row_begin || i >= row_end) continue;
            const int64_t q = fill[pos[i]]++;
            col[q] = (uint32_t)j;
            wt[q] = w ? w[e] : 1.0f;
        }
    }
    // 2. levels: row i must run after every neighbour j (j>i) that the reference visits earlier.
    std::vector<int32_t> level(nrows, 0);
    int32_t nlevels = nrows ? 1 : 0;
    for (int64_t r = 0; r < nrows; ++r) {
        int32_t lv = 0;
        for (int64_t q = off[r]; q < off[r + 1]; ++q) {
            const int32_t pj = pos[col[q]];
            if (pj >= 0 && pj < r) {          // j already updated in this sweep -> read X_new[j]
                col[q] |= 0x80000000u;
                lv = std::max(lv, level[pj] + 1);
            }
        }
        level[r] = lv;
        nlevels = std::max(nlevels, lv + 1);
    }
    // 3. stable sort rows by level
    std::vector<int64_t> lvl_cnt(nlevels + 1, 0);
    for (int64_t r = 0; r < nrows; ++r) ++lvl_cnt[level[r] + 1];
    for (int32_t l = 0; l < nlevels; ++l) lvl_cnt[l + 1] += lvl_cnt[l];
    std::vector<int64_t> level_hubs;
    std::vector<int32_t> rows_sorted(nrows);
    std::vector<int64_t> ptr_sorted(nrows + 1, 0);
    std::vector<uint32_t> col_sorted(nupd);
    std::vector<float> w_sorted(nupd);
    {
        // inside a level the rows are independent: hub rows (gf_hub_kernel) first, the others keep the reference's visiting order
        std::vector<int64_t> at(lvl_cnt.begin(), lvl_cnt.end() - 1);
        std::vector<int64_t> newpos(nrows);
        level_hubs.assign(nlevels, 0);
        const bool hubs_on = getenv("GEMHIP_GF_NO_HUB_KERNEL") == nullptr;
        const int64_t hub_t = getenv("GEMHIP_GF_HUB_EDGES") ? std::max(1, atoi(getenv("GEMHIP_GF_HUB_EDGES"))) : GF_HUB_EDGES;   // (tests lower it)
        for (int64_t r = 0; r < nrows; ++r) if (hubs_on && off[r + 1] - off[r] >= hub_t) { newpos[r] = at[level[r]]++; ++level_hubs[level[r]]; }
        for (int64_t r = 0; r < nrows; ++r) if (!(hubs_on && off[r + 1] - off[r] >= hub_t)) newpos[r] = at[level[r]]++;
        std::vector<int64_t> inv(nrows);
        for (int64_t r = 0; r < nrows; ++r) inv[newpos[r]] = r;
        for (int64_t s = 0; s < nrows; ++s) {
            const int64_t r = inv[s];
            rows_sorted[s] = order[r];
            ptr_sorted[s + 1] = ptr_sorted[s] + (off[r + 1] - off[r]);
            std::copy(col.begin() + off[r], col.begin() + off[r + 1], col_sorted.begin() + ptr_sorted[s]);
            std::copy(wt.begin() + off[r], wt.begin() + off[r + 1], w_sorted.begin() + ptr_sorted[s]);
        }
    }

    auto *p = new gemhip_gf_plan();
    p->n = n; p->d = d; p->nrows = nrows; p->nupd = nupd;
    p->level_off.assign(lvl_cnt.begin(), lvl_cnt.end());
    p->level_hubs = level_hubs;
    p->level_maxlen.assign(nlevels, 0);
    for (int32_t l = 0; l < nlevels; ++l)
        for (int64_t q = lvl_cnt[l] + level_hubs[l]; q < lvl_cnt[l + 1]; ++q) p->level_maxlen[l] = std::max(p->level_maxlen[l], ptr_sorted[q + 1] - ptr_sorted[q]);
    if (hipGetDevice(&p->device) != hipSuccess) { delete p; return fail(GEMHIP_E_HIP, "gf_plan_create: no HIP device"); }
    phase_acc()[PH_HOST] += phase_now() - t_host0;
    PhaseScope ph_up(PH_H2D);
    auto up = [&](void **dp, const void *hp, size_t bytes) -> hipError_t {
        hipError_t e = hipMalloc(dp, bytes ? bytes : 16);
        if (e != hipSuccess) return e;
        return bytes ? hipMemcpy(*dp, hp, bytes, hipMemcpyHostToDevice) : hipSuccess;
    };
    hipError_t e = up((void **)&p->d_rows, rows_sorted.data(), nrows * sizeof(int32_t));
    if (e == hipSuccess) e = up((void **)&p->d_ptr, ptr_sorted.data(), (nrows + 1) * sizeof(int64_t));
    if (e == hipSuccess) e = up((void **)&p->d_col, col_sorted.data(), nupd * sizeof(uint32_t));
    if (e == hipSuccess) e = up((void **)&p->d_w, w_sorted.data(), nupd * sizeof(float));
    if (e != hipSuccess) {
        gemhip_gf_plan_destroy(p);
        return fail(GEMHIP_E_HIP, "gf_plan_create: device upload failed: %s", hipGetErrorString(e));
    }
    *out = p;
    return GEMHIP_OK;
}

extern "C" int gemhip_gf_plan_destroy(gemhip_gf_plan_t p)
{
    if (!p) return GEMHIP_OK;
    hipFree(p->d_rows); hipFree(p->d_ptr); hipFree(p->d_col); hipFree(p->d_w);
    if (p->hub_stream) { hipStreamSynchronize(p->hub_stream); hipStreamDestroy(p->hub_stream); hipEventDestroy(p->hub_fork); hipEventDestroy(p->hub_join); }
    if (p->own_X) { hipFree(p->X[0]); hipFree(p->X[1]); }
    delete p;
    return GEMHIP_OK;
}

static int ensure_tables(gemhip_gf_plan_t p)
{
    if (p->X[0]) return GEMHIP_OK;
    const size_t bytes = (size_t)p->n * p->d * sizeof(float);
    GEMHIP_CHECK(hipMalloc((void **)&p->X[0], bytes));
    GEMHIP_CHECK(hipMalloc((void **)&p->X[1], bytes));
    p->own_X = true;
    p->cur = 0;
    return GEMHIP_OK;
}

extern "C" int gemhip_gf_plan_bind(gemhip_gf_plan_t p, void *dXa, void *dXb)
{
    GEMHIP_REQUIRE(p && dXa && dXb && dXa != dXb, "gf_plan_bind: need two distinct device buffers");
    if (p->own_X) { hipFree(p->X[0]); hipFree(p->X[1]); p->own_X = false; }
    p->X[0] = (float *)dXa; p->X[1] = (float *)dXb; p->cur = 0;
    return GEMHIP_OK;
}

extern "C" int gemhip_gf_plan_set_embedding(gemhip_gf_plan_t p, const float *X_host)
{
    GEMHIP_REQUIRE(p && X_host, "gf_plan_set_embedding: NULL argument");
    if (int rc = ensure_tables(p)) return rc;
    const size_t bytes = (size_t)p->n * p->d * sizeof(float);
    PhaseScope ph(PH_H2D);
    GEMHIP_CHECK(hipMemcpy(p->X[0], X_host, bytes, hipMemcpyHostToDevice));
    GEMHIP_CHECK(hipMemcpy(p->X[1], p->X[0], bytes, hipMemcpyDeviceToDevice));
    p->cur = 0;
    return GEMHIP_OK;
}

extern "C" int gemhip_gf_plan_init_embedding(gemhip_gf_plan_t p, uint64_t seed, float scale)
{
    GEMHIP_REQUIRE(p, "gf_plan_init_embedding: NULL plan");
    if (int rc = ensure_tables(p)) return rc;
    const int64_t total = p->n * p->d;
    const int64_t threads = (total + 3) / 4;
    hipLaunchKernelGGL(gf_init_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, 0, p->X[0], total, seed, scale);
    GEMHIP_CHECK(hipGetLastError());
    GEMHIP_CHECK(hipMemcpy(p->X[1], p->X[0], (size_t)total * sizeof(float), hipMemcpyDeviceToDevice));
    p->cur = 0;
    return GEMHIP_OK;
}

extern "C" int gemhip_gf_plan_sweeps(gemhip_gf_plan_t p, int32_t nsweeps, float eta, float regu, void *stream)
{
    GEMHIP_REQUIRE(p && nsweeps >= 0, "gf_plan_sweeps: bad arguments");
    GEMHIP_REQUIRE(p->X[0] && p->X[1], "gf_plan_sweeps: no embedding table (call set/init/bind first)");
    if (p->nrows == 0) return GEMHIP_OK;
    const sweep_fn fn = pick_sweep((int)p->d);
    hipStream_t s = (hipStream_t)stream;
    const int nlevels = (int)p->level_off.size() - 1;
    int it0 = 0;
    if (p->fused_sweeps > 1 && nlevels == 1 && (p->level_hubs.empty() || p->level_hubs[0] == 0) && pick_coop((int)p->d)) {
        // up to fused_sweeps sweeps per cooperative launch; the table of sweep k is X[cur ^ (k & 1)]
        while (nsweeps - it0 >= 2) {
            const int k = std::min(nsweeps - it0, p->fused_sweeps);
            if (int rc = pick_coop((int)p->d)(p, p->X[p->cur], p->X[p->cur ^ 1], eta, regu, k, s)) return rc;
            if (k & 1) p->cur ^= 1;
            it0 += k;
        }
    }
    for (int it = it0; it < nsweeps; ++it) {
        const float *Xold = p->X[p->cur];
        float *Xnew = p->X[p->cur ^ 1];
        for (int l = 0; l < nlevels; ++l) {
            const int64_t r0 = p->level_off[l], nr = p->level_off[l + 1] - r0;
            const int64_t nh = l < (int)p->level_hubs.size() ? p->level_hubs[l] : 0;
            if (nh > 0) {                              // hub rows of this level on a side stream, next to the wave-per-row kernel
                if (!p->hub_stream) {
                    GEMHIP_CHECK(hipStreamCreateWithFlags(&p->hub_stream, hipStreamNonBlocking));
                    GEMHIP_CHECK(hipEventCreateWithFlags(&p->hub_fork, hipEventDisableTiming));
                    GEMHIP_CHECK(hipEventCreateWithFlags(&p->hub_join, hipEventDisableTiming));
                }
                GEMHIP_CHECK(hipEventRecord(p->hub_fork, s));
                GEMHIP_CHECK(hipStreamWaitEvent(p->hub_stream, p->hub_fork, 0));
                pick_hub((int)p->d)(p, r0, nh, Xold, Xnew, eta, regu, p->hub_stream);
                GEMHIP_CHECK(hipEventRecord(p->hub_join, p->hub_stream));
            }
            if (nr - nh > 0) fn(p, r0 + nh, nr - nh, Xold, Xnew, eta, regu, s);
            if (nh > 0) GEMHIP_CHECK(hipStreamWaitEvent(s, p->hub_join, 0));
        }
        p->cur ^= 1;
    }
    GEMHIP_CHECK(hipGetLastError());
    return GEMHIP_OK;
}

extern "C" int gemhip_gf_plan_set_rows_per_wave(gemhip_gf_plan_t p, int32_t rows_per_wave)
{
    GEMHIP_REQUIRE(p && rows_per_wave >= 0 && rows_per_wave <= 64, "gf_plan_set_rows_per_wave: 0 (auto) .. 64");
    p->rows_per_wave = rows_per_wave;
    return GEMHIP_OK;
}

extern "C" int gemhip_gf_plan_set_fused_sweeps(gemhip_gf_plan_t p, int32_t sweeps_per_launch, int32_t max_grid)
{
    GEMHIP_REQUIRE(p && sweeps_per_launch >= 0 && sweeps_per_launch <= 65536 && max_grid >= 0, "gf_plan_set_fused_sweeps: bad arguments");
    p->fused_sweeps = sweeps_per_launch; p->fused_grid = max_grid;
    return GEMHIP_OK;
}

extern "C" int gemhip_gf_plan_get_embedding(gemhip_gf_plan_t p, float *X_host)
{
    GEMHIP_REQUIRE(p && X_host && p->X[0], "gf_plan_get_embedding: bad arguments");
    GEMHIP_CHECK(hipDeviceSynchronize());
    PhaseScope ph(PH_D2H);
    GEMHIP_CHECK(hipMemcpy(X_host, p->X[p->cur], (size_t)p->n * p->d * sizeof(float), hipMemcpyDeviceToHost));
    return GEMHIP_OK;
}

extern "C" int gemhip_gf_plan_current(gemhip_gf_plan_t p, void **dX)
{
    GEMHIP_REQUIRE(p && dX, "gf_plan_current: NULL argument");
    *dX = p->X[p->cur];
    return GEMHIP_OK;
}

extern "C" int gemhip_gf_plan_info(gemhip_gf_plan_t p, int64_t *info)
{
    GEMHIP_REQUIRE(p && info, "gf_plan_info: NULL argument");
    info[0] = p->nupd; info[1] = p->nrows; info[2] = (int64_t)p->level_off.size() - 1; info[3] = p->n; info[4] = p->d;
    // 3*4d + 12 bytes per update (read X_i, read X_j, write X_i, (i,j,w))
    info[5] = p->nupd * (3 * 4 * p->d + 12);
    {   // rows per wavefront the largest level's sweep launch will use (1: gf_sweep_kernel, > 1: gf_sweep_rows_kernel)
        int64_t big = 0;
        for (size_t l = 0; l + 1 < p->level_off.size(); ++l) big = std::max<int64_t>(big, p->level_off[l + 1] - p->level_off[l]);
        int64_t bigmax = 0;
        for (size_t l = 0; l + 1 < p->level_off.size(); ++l)
            if (p->level_off[l + 1] - p->level_off[l] == big) bigmax = p->level_maxlen.size() > l ? p->level_maxlen[l] : 0;
        info[6] = p->rows_per_wave > 0 ? p->rows_per_wave : (bigmax > 2 * WAVE ? 1 : gf_rows_per_wave(big));
    }
    info[7] = 0;
    return GEMHIP_OK;
}

extern "C" int gemhip_gf_train(int64_t n, int64_t m, const int32_t *src, const int32_t *dst, const float *w, int32_t d, float eta,
                               float regu, int32_t max_iter, float *X_inout, double *stats)
{
    GEMHIP_REQUIRE(X_inout != nullptr, "gf_train: X_inout is NULL");
    GEMHIP_REQUIRE(max_iter >= 0, "gf_train: max_iter=%d", max_iter);
    for (int k = 0; k < PH_COUNT; ++k) phase_acc()[k] = 0.0;
    const double t_call = phase_now();
    gemhip_gf_plan_t p = nullptr;
    int rc = gemhip_gf_plan_create(n, m, src, dst, w, d, 0, n, &p);
    if (rc) return rc;
    rc = gemhip_gf_plan_set_embedding(p, X_inout);
    hipEvent_t t0 = nullptr, t1 = nullptr;
    float ms = 0.f;
    if (!rc && (hipEventCreate(&t0) != hipSuccess || hipEventCreate(&t1) != hipSuccess)) rc = fail(GEMHIP_E_HIP, "gf_train: hipEventCreate");
    if (!rc) {
        hipEventRecord(t0, 0);
        rc = gemhip_gf_plan_sweeps(p, max_iter, eta, regu, nullptr);
        hipEventRecord(t1, 0);
    }
    if (!rc) rc = gemhip_gf_plan_get_embedding(p, X_inout);
    if (!rc) hipEventElapsedTime(&ms, t0, t1);
    phase_acc()[PH_KERNELS] = ms * 1e-3;
    if (stats && !rc) {
        stats[0] = ms * 1e-3; stats[1] = (double)p->nupd; stats[2] = (double)p->nrows;
        stats[3] = (double)(p->level_off.size() - 1);
    }
    if (t0) hipEventDestroy(t0);
    if (t1) hipEventDestroy(t1);
    gemhip_gf_plan_destroy(p);
    phase_acc()[PH_TOTAL] = phase_now() - t_call;
    return rc;
}

extern "C" int gemhip_gf_objective(int64_t n, int64_t m, const int32_t *src, const int32_t *dst, const float *w, int32_t d,
                                   const float *X_host, double *out)
{
    GEMHIP_REQUIRE(n > 0 && m >= 0 && d >= 1 && X_host && out, "gf_objective: bad arguments");
    int32_t *ds = nullptr, *dd = nullptr; float *dw = nullptr, *dX = nullptr; double *dout = nullptr;
    int rc = GEMHIP_OK;
    auto cleanup = [&]() { hipFree(ds); hipFree(dd); hipFree(dw); hipFree(dX); hipFree(dout); };
#define OBJ_TRY(x) do { hipError_t _e = (x); if (_e != hipSuccess) { cleanup(); return fail(GEMHIP_E_HIP, "gf_objective: %s: %s", #x, hipGetErrorString(_e)); } } while (0)
    OBJ_TRY(hipMalloc((void **)&ds, std::max<int64_t>(m, 1) * 4));
    OBJ_TRY(hipMalloc((void **)&dd, std::max<int64_t>(m, 1) * 4));
    if (w) OBJ_TRY(hipMalloc((void **)&dw, std::max<int64_t>(m, 1) * 4));
    OBJ_TRY(hipMalloc((void **)&dX, (size_t)n * d * 4));
    OBJ_TRY(hipMalloc((void **)&dout, 16));
    if (m) {
        OBJ_TRY(hipMemcpy(ds, src, m * 4, hipMemcpyHostToDevice));
        OBJ_TRY(hipMemcpy(dd, dst, m * 4, hipMemcpyHostToDevice));
        if (w) OBJ_TRY(hipMemcpy(dw, w, m * 4, hipMemcpyHostToDevice));
    }
    OBJ_TRY(hipMemcpy(dX, X_host, (size_t)n * d * 4, hipMemcpyHostToDevice));
    OBJ_TRY(hipMemset(dout, 0, 16));
    hipLaunchKernelGGL(gf_objective_kernel, dim3(2048), dim3(256), 0, 0, ds, dd, dw, dX, m, n, (int)d, dout);
    OBJ_TRY(hipGetLastError());
    OBJ_TRY(hipMemcpy(out, dout, 16, hipMemcpyDeviceToHost));
#undef OBJ_TRY
    cleanup();
    return rc;
}
