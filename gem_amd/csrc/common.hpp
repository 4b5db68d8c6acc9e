// common.hpp -- shared host/device helpers for libgem_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string>
#include <cstdio>
#include <cstdarg>

#include "../../include/gem_hip.h"

namespace gemhip {

// ---------------------------------------------------------------- error state
std::string &last_error_ref();
int fail(int code, const char *fmt, ...);

#define GEMHIP_CHECK(expr)                                                        \
    do {                                                                          \
        hipError_t _e = (expr);                                                   \
        if (_e != hipSuccess)                                                     \
            return ::gemhip::fail(GEMHIP_E_HIP, "%s failed: %s (%s:%d)", #expr,   \
                                  hipGetErrorString(_e), __FILE__, __LINE__);     \
    } while (0)

#define GEMHIP_REQUIRE(cond, ...)                                  \
    do {                                                           \
        if (!(cond))                                               \
            return ::gemhip::fail(GEMHIP_E_INVALID, __VA_ARGS__);  \
    } while (0)

// ---------------------------------------------------------------- where a one-shot call's wall time went (gemhip_last_call_phases)
// Thread-local accumulators the one-shot entry points (gemhip_gf_train, gemhip_n2v_train, gemhip_hope) reset on entry; the staged calls they
// are made of add host-side preparation (sorting, Vose tables), host->device and device->host copy time to them.  Wall-clock, host side.
enum { PH_TOTAL = 0, PH_HOST = 1, PH_H2D = 2, PH_KERNELS = 3, PH_D2H = 4, PH_COUNT = 8 };
double *phase_acc();
double phase_now();
struct PhaseScope {
    int k; double t0;
    explicit PhaseScope(int kk) : k(kk), t0(phase_now()) {}
    ~PhaseScope() { phase_acc()[k] += phase_now() - t0; }
};

constexpr int WAVE = 64;          // CDNA wavefront
constexpr int NUM_XCD = 8;        // MI355X: 8 XCDs, block b lands on XCD b % 8

// ---------------------------------------------------------------- device side
#if defined(__HIPCC__)

// Wave-wide fp32 sum, result in every lane.  4 DPP steps reduce inside each
// row of 16 lanes (no LDS traffic), then 4 v_readlane fold the four rows.
__device__ __forceinline__ float wave_sum(float v)
{
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));  // quad_perm [1,0,3,2]
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));  // quad_perm [2,3,0,1]
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xF, 0xF, true)); // row_half_mirror
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xF, 0xF, true)); // row_mirror
    const int iv = __builtin_bit_cast(int, v);
    const float r0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(iv, 0));
    const float r1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(iv, 16));
    const float r2 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(iv, 32));
    const float r3 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(iv, 48));
    return (r0 + r1) + (r2 + r3);
}

__device__ __forceinline__ int lane_id() { return threadIdx.x & (WAVE - 1); }

template <typename T>
__device__ __forceinline__ T bcast_lane(T v, int srclane)   // srclane must be wave-uniform
{
    static_assert(sizeof(T) == 4, "32-bit only");
    return __builtin_bit_cast(T, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), srclane));
}

// XCD-aware work mapping: hardware places block b on XCD b % 8.  Give every XCD
// one CONTIGUOUS eighth of the work-item range so that neighbouring rows (which
// share most of their gather targets in block-structured graphs) meet in the
// same 4 MiB L2.  Speed only -- correctness never depends on placement.
__device__ __forceinline__ int64_t xcd_contiguous_block(int64_t b, int64_t nblocks)
{
    const int64_t per = (nblocks + NUM_XCD - 1) / NUM_XCD;
    const int64_t xcd = b % NUM_XCD, idx = b / NUM_XCD;
    return xcd * per + idx;        // may be >= nblocks for the ragged tail: caller checks
}

// ------------------------------------------------------------- Philox4x32-10
// Counter-based RNG (Salmon et al. 2011).  Stateless: every draw is a pure
// function of (seed, counter), so the CPU oracle reproduces the device stream
// exactly and results do not depend on scheduling.
struct u32x4 { uint32_t x, y, z, w; };

__host__ __device__ __forceinline__ u32x4 philox4x32_10(uint64_t seed, uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3)
{
    uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c0;
        const uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
        const uint32_t n1 = (uint32_t)p1;
        const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        const uint32_t n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    return u32x4{c0, c1, c2, c3};
}

// uniform in [0,1) with 24 bits, exactly representable in fp32
__host__ __device__ __forceinline__ float u01(uint32_t r) { return (float)(r >> 8) * (1.0f / 16777216.0f); }

#endif  // __HIPCC__

}  // namespace gemhip
