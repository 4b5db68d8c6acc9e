// A/B of the three mechanisms where the kernels depart from what BASELINE.json's north_star sketches (VERDICT r5 "weak" #9), one JSON line each:
//   (1) "warp-shuffle negative-sample dot products": six 128-term dot products reduced with ds_bpermute shuffles (6 x 6 __shfl_xor steps) against the
//       kernels' wave_sum6 (DPP quad_perm / row_ror + v_permlane{16,32}_swap, transposing while reducing: 22 lane operations for all six sums);
//   (2) "CSR adjacency staged through LDS": one wavefront walks a CSR row -- 64 (col, w) pairs loaded coalesced and staged through LDS (write, barrier,
//       broadcast reads) against register load + v_readlane broadcast (what gf_sweep_kernel / hope_spmm_kernel do), gathering a 512-byte row per neighbour;
//   (3) "per-wavefront alias-table walk sampling": a wavefront per walker (lane 0 walks, 63 lanes idle) against a lane per walker (n2v_walk_kernel), uniform
//       first-order steps on a CSR graph, same number of walkers and steps.
//   hipcc --offload-arch=gfx950 -O3 -o scripts/microbench/northstar scripts/microbench/northstar.hip && scripts/microbench/northstar
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#include <vector>
#include <random>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

__device__ __forceinline__ uint32_t hash32(uint32_t h) { h ^= h >> 16; h *= 0x85EBCA6Bu; h ^= h >> 13; h *= 0xC2B2AE35u; h ^= h >> 16; return h; }

// ---------------------------------------------------------------- (1) six wave sums
template <int CTRL>
__device__ __forceinline__ float dpp_mov(float v) { return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true)); }
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float wave_sum6_dpp(const float (&p)[6], int lane)      // (the kernels' reduction, sgns.hpp)
{
    const bool b0 = (lane & 1) != 0, b1 = (lane & 2) != 0, b2 = (lane & 4) != 0;
    const float a0 = (b0 ? p[1] : p[0]) + dpp_mov<0xB1>(b0 ? p[0] : p[1]);
    const float a1 = (b0 ? p[3] : p[2]) + dpp_mov<0xB1>(b0 ? p[2] : p[3]);
    const float a2 = (b0 ? p[5] : p[4]) + dpp_mov<0xB1>(b0 ? p[4] : p[5]);
    float c0 = (b1 ? a1 : a0) + dpp_mov<0x4E>(b1 ? a0 : a1);
    float c1 = a2 + dpp_mov<0x4E>(a2);
    c0 += dpp_mov<0x124>(c0); c0 += dpp_mov<0x128>(c0);
    c1 += dpp_mov<0x124>(c1); c1 += dpp_mov<0x128>(c1);
    float m = b2 ? c1 : c0;
    u32x2 r = __builtin_amdgcn_permlane16_swap(__builtin_bit_cast(unsigned, m), __builtin_bit_cast(unsigned, m), false, false);
    unsigned lo = r.x, hi = r.y;
    m = __builtin_bit_cast(float, lo) + __builtin_bit_cast(float, hi);
    r = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(unsigned, m), __builtin_bit_cast(unsigned, m), false, false);
    lo = r.x; hi = r.y;
    return __builtin_bit_cast(float, lo) + __builtin_bit_cast(float, hi);
}
__device__ __forceinline__ float wave_sum_shfl(float v)          // butterfly of ds_bpermute shuffles: every lane ends with the total
{
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
template <bool DPP>
__global__ __launch_bounds__(64) void sum6_kernel(int iters, float *sink)
{
    const int lane = threadIdx.x;
    float x[2], y[6][2];
    x[0] = 1.0f + lane * 1e-3f; x[1] = 0.5f - lane * 1e-3f;
#pragma unroll
    for (int j = 0; j < 6; ++j) { y[j][0] = 0.1f * (j + 1) + lane * 1e-4f; y[j][1] = -0.05f * (j + 1); }
    float acc = 0.f;
    for (int it = 0; it < iters; ++it) {
        float part[6];
#pragma unroll
        for (int j = 0; j < 6; ++j) part[j] = fmaf(x[0], y[j][0], x[1] * y[j][1]);
        float g[6];
        if constexpr (DPP) {
            const float f = wave_sum6_dpp(part, lane);           // lanes 0..5 hold totals 0..5
#pragma unroll
            for (int j = 0; j < 6; ++j) g[j] = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, f), j));
        } else {
#pragma unroll
            for (int j = 0; j < 6; ++j) g[j] = wave_sum_shfl(part[j]);
        }
#pragma unroll
        for (int j = 0; j < 6; ++j) { y[j][0] = fmaf(1e-6f * g[j], x[0], y[j][0]); acc += g[j]; }     // (the update the gradient feeds: a dependent chain like the kernel's)
        x[0] += 1e-7f * acc;
    }
    if (lane == 0) sink[blockIdx.x] = acc + y[0][0];
}

// ---------------------------------------------------------------- (2) CSR row: LDS staging vs register + readlane
template <bool LDS>
__global__ __launch_bounds__(256) void csr_kernel(int64_t nrows, const int64_t *__restrict__ row_ptr, const int32_t *__restrict__ col, const float *__restrict__ w,
                                                  const float *__restrict__ X, float *__restrict__ Y)
{
    __shared__ int32_t s_col[4][64];
    __shared__ float s_w[4][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t i = (int64_t)blockIdx.x * 4 + wave;
    if (i >= nrows) return;
    float2 acc = make_float2(0.f, 0.f);
    for (int64_t e = row_ptr[i]; e < row_ptr[i + 1]; e += 64) {
        const int cnt = (int)((row_ptr[i + 1] - e) < 64 ? (row_ptr[i + 1] - e) : 64);
        const int32_t cj = lane < cnt ? col[e + lane] : 0;
        const float wj = lane < cnt ? w[e + lane] : 0.f;
        if constexpr (LDS) { s_col[wave][lane] = cj; s_w[wave][lane] = wj; __builtin_amdgcn_wave_barrier(); }
        for (int k = 0; k < cnt; k += 4) {
            float2 xr[4]; float ww[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int kk = (k + u) < cnt ? (k + u) : (cnt - 1);
                int32_t c; float v;
                if constexpr (LDS) { c = s_col[wave][kk]; v = s_w[wave][kk]; }
                else { c = __builtin_amdgcn_readlane(cj, kk); v = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, wj), kk)); }
                ww[u] = (k + u) < cnt ? v : 0.f;
                xr[u] = *reinterpret_cast<const float2 *>(X + (int64_t)c * 128 + lane * 2);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) { acc.x = fmaf(ww[u], xr[u].x, acc.x); acc.y = fmaf(ww[u], xr[u].y, acc.y); }
        }
        if constexpr (LDS) __builtin_amdgcn_wave_barrier();
    }
    *reinterpret_cast<float2 *>(Y + i * 128 + lane * 2) = acc;
}

// ---------------------------------------------------------------- (3) walkers: one per lane vs one per wavefront
template <bool PER_WAVE>
__global__ __launch_bounds__(256) void walk_kernel(int64_t nwalkers, int steps, const int64_t *__restrict__ row_ptr, const int32_t *__restrict__ col, int32_t nnodes,
                                                   int32_t *__restrict__ last)
{
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t walker = PER_WAVE ? t / 64 : t;
    if (walker >= nwalkers) return;
    if (PER_WAVE && (threadIdx.x & 63) != 0) return;           // 63 lanes idle: the walk is one dependent chain
    int32_t v = (int32_t)(hash32((uint32_t)walker) % (uint32_t)nnodes);
    for (int s = 0; s < steps; ++s) {
        const int64_t a = row_ptr[v], b = row_ptr[v + 1];
        if (b == a) break;
        const uint32_t r = hash32((uint32_t)walker * 0x9E3779B9u + (uint32_t)s);
        v = col[a + (int64_t)(((uint64_t)r * (uint64_t)(b - a)) >> 32)];
    }
    last[walker] = v;
}

template <class F>
static double time_ms(F f, int reps = 3)
{
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    f(); CK(hipDeviceSynchronize());
    float best = 1e30f;
    for (int r = 0; r < reps; ++r) { CK(hipEventRecord(e0)); f(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); float ms; CK(hipEventElapsedTime(&ms, e0, e1)); best = ms < best ? ms : best; }
    return best;
}

int main()
{
    float *sink; CK(hipMalloc(&sink, 1 << 20));
    {   // (1) 1792 wavefronts (the SGNS launch), 20 000 pair steps each
        const int waves = 1792, iters = 20000;
        const double a = time_ms([&] { hipLaunchKernelGGL((sum6_kernel<true>), dim3(waves), dim3(64), 0, 0, iters, sink); });
        const double b = time_ms([&] { hipLaunchKernelGGL((sum6_kernel<false>), dim3(waves), dim3(64), 0, 0, iters, sink); });
        printf("{\"ab\": \"six wave sums per pair step\", \"dpp_wave_sum6_ns_per_step\": %.2f, \"ds_bpermute_shuffles_ns_per_step\": %.2f, \"ratio\": %.2f, \"waves\": %d}\n",
               a * 1e6 / iters, b * 1e6 / iters, b / a, waves);
    }
    {   // (2) SBM-like CSR: 1M rows, 10 neighbours each (uniformly random), 512-byte rows
        const int64_t n = 1000000, deg = 10;
        std::vector<int64_t> rp(n + 1); std::vector<int32_t> cl(n * deg); std::vector<float> ww(n * deg, 0.5f);
        std::mt19937 rng(1);
        for (int64_t i = 0; i <= n; ++i) rp[i] = i * deg;
        for (auto &c : cl) c = (int32_t)(rng() % n);
        int64_t *d_rp; int32_t *d_cl; float *d_w, *X, *Y;
        CK(hipMalloc(&d_rp, (n + 1) * 8)); CK(hipMalloc(&d_cl, n * deg * 4)); CK(hipMalloc(&d_w, n * deg * 4)); CK(hipMalloc(&X, n * 512)); CK(hipMalloc(&Y, n * 512));
        CK(hipMemcpy(d_rp, rp.data(), (n + 1) * 8, hipMemcpyHostToDevice)); CK(hipMemcpy(d_cl, cl.data(), n * deg * 4, hipMemcpyHostToDevice));
        CK(hipMemcpy(d_w, ww.data(), n * deg * 4, hipMemcpyHostToDevice)); CK(hipMemset(X, 0, n * 512));
        const double a = time_ms([&] { hipLaunchKernelGGL((csr_kernel<false>), dim3((unsigned)((n + 3) / 4)), dim3(256), 0, 0, n, d_rp, d_cl, d_w, X, Y); });
        const double b = time_ms([&] { hipLaunchKernelGGL((csr_kernel<true>), dim3((unsigned)((n + 3) / 4)), dim3(256), 0, 0, n, d_rp, d_cl, d_w, X, Y); });
        printf("{\"ab\": \"CSR row of one wavefront: (col, w) by register load + v_readlane vs staged through LDS\", \"readlane_ms\": %.3f, \"lds_ms\": %.3f, \"ratio\": %.3f, "
               "\"rows\": %lld, \"neighbours_per_row\": %lld}\n", a, b, b / a, (long long)n, (long long)deg);
        // (3) on the same graph: 1M walkers x 80 steps
        int32_t *last; CK(hipMalloc(&last, n * 4));
        const int steps = 80;
        const double c = time_ms([&] { hipLaunchKernelGGL((walk_kernel<false>), dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, n, steps, d_rp, d_cl, (int32_t)n, last); });
        const double d = time_ms([&] { hipLaunchKernelGGL((walk_kernel<true>), dim3((unsigned)((n * 64 + 255) / 256)), dim3(256), 0, 0, n, steps, d_rp, d_cl, (int32_t)n, last); }, 1);
        printf("{\"ab\": \"walkers: one per lane vs one per wavefront\", \"lane_per_walker_ms\": %.2f, \"wavefront_per_walker_ms\": %.2f, \"ratio\": %.1f, \"walkers\": %lld, \"steps\": %d}\n",
               c, d, d / c, (long long)n, steps);
    }
    return 0;
}
