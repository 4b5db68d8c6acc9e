// eval.hip -- graph-reconstruction average precision of sampled nodes on the GPU.
//
// Reference: evaluateStaticGraphReconstruction (gem/evaluation/evaluate_graph_reconstruction.py:8-46) builds
// the n x n matrix of get_edge_weight(i, j) with an O(n^2) Python loop (static_graph_embedding.py:60-64), lists
// the pairs i<j with weight > 0 (gem/utils/evaluation_util.py:28-35), sorts every node's candidates by weight
// (stable, descending) and averages precision at the true edges (gem/evaluation/metrics.py:6-46).  That is
// unusable beyond n ~ 1e4, yet MAP is the parity metric of this backend at 1M nodes.
//
// AP of node i does not need a sort: for every true neighbour t,
//     rank_all(t) = 1 + #{candidates j : s_j > s_t  or (s_j == s_t and j < t)}      (stable tie rule)
//     rank_hit(t) = the same count restricted to true neighbours
//     AP_i = (1/H) sum_t rank_hit(t) / rank_all(t)       over the H neighbours with s_t > 0.
// One workgroup per sampled node streams all rows B_j once (a wavefront per row, lanes across the d columns, fp64
// dot of the fp32 inputs so ranks agree with the float64 reference), and every lane compares the score of the row
// with "its" neighbours' scores -- O(n d) per node, exact, no n x n matrix.
// score(i, j) = A_i . B_j : A = B = X for GF / node2vec (X_i . X_j), A = X[:, :k], B = X[:, k:] for HOPE (hope.py:43-44).
#include "common.hpp"
#include <vector>

using namespace gemhip;

namespace {

constexpr int EV_BLOCK = 256;
constexpr int EV_MAXNB = 512;                 // true neighbours per sampled node handled in registers
constexpr int EV_NREG = EV_MAXNB / WAVE;

__device__ __forceinline__ double wave_sum_f64(double v)
{
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

template <int NV>     // lane l holds columns (c*64 + l), c < NV  (da <= 64*NV)
__global__ __launch_bounds__(EV_BLOCK) void eval_ap_kernel(int64_t n, int da, const float *__restrict__ A, const float *__restrict__ B, int ldb,
                                                           const int64_t *__restrict__ row_ptr, const int32_t *__restrict__ col, int undirected,
                                                           const int32_t *__restrict__ nodes, double *__restrict__ ap_out, int *__restrict__ err)
{
    __shared__ double s_nb[EV_MAXNB];
    __shared__ int t_nb[EV_MAXNB];
    __shared__ int cnt_nb[EV_MAXNB];
    __shared__ int nnb_s;
    const int lane = lane_id(), wave = threadIdx.x >> 6;
    const int i = nodes[blockIdx.x];
    // ---- true neighbours that are candidates (j != i, and j > i when undirected)
    if (threadIdx.x == 0) {
        int c = 0; bool overflow = false;
        for (int64_t e = row_ptr[i]; e < row_ptr[i + 1]; ++e) {
            const int t = col[e];
            if (t == i || (undirected && t < i)) continue;
            if (c < EV_MAXNB) t_nb[c++] = t; else overflow = true;
        }
        nnb_s = c;
        if (overflow) atomicExch(err, 1);
    }
    for (int k = threadIdx.x; k < EV_MAXNB; k += EV_BLOCK) cnt_nb[k] = 0;
    __syncthreads();
    const int nnb = nnb_s;
    float a[NV];
#pragma unroll
    for (int c = 0; c < NV; ++c) { const int cc = c * WAVE + lane; a[c] = cc < da ? A[(int64_t)i * ldb + cc] : 0.f; }
    auto score = [&](int64_t j) -> double {
        double part = 0.0;
        const float *bj = B + j * ldb;
#pragma unroll
        for (int c = 0; c < NV; ++c) { const int cc = c * WAVE + lane; if (cc < da) part += (double)a[c] * (double)bj[cc]; }
        return wave_sum_f64(part);
    };
    for (int k = wave; k < nnb; k += EV_BLOCK / WAVE) {
        const double s = score(t_nb[k]);
        if (lane == 0) s_nb[k] = s;
    }
    __syncthreads();
    double my_s[EV_NREG]; int my_t[EV_NREG]; int my_c[EV_NREG];
#pragma unroll
    for (int r = 0; r < EV_NREG; ++r) {
        const int k = r * WAVE + lane;
        my_s[r] = k < nnb ? s_nb[k] : 0.0; my_t[r] = k < nnb ? t_nb[k] : -1; my_c[r] = 0;
    }
    // ---- stream the candidates
    const int64_t lo = undirected ? (int64_t)i + 1 : 0;
    const int nreg_used = (nnb + WAVE - 1) / WAVE;
    if (nnb > 0) {
        for (int64_t j = lo + wave; j < n; j += EV_BLOCK / WAVE) {
            if (j == i) continue;
            const double s = score(j);
            if (!(s > 0.0)) continue;                                  // evaluation_util.py:34  adj[i, j] > threshold (0.0)
#pragma unroll
            for (int r = 0; r < EV_NREG; ++r)
                if (r < nreg_used) my_c[r] += (s > my_s[r] || (s == my_s[r] && j < my_t[r])) ? 1 : 0;
        }
#pragma unroll
        for (int r = 0; r < EV_NREG; ++r) {
            const int k = r * WAVE + lane;
            if (k < nnb && my_c[r]) atomicAdd(&cnt_nb[k], my_c[r]);
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        double sum = 0.0; int H = 0;
        for (int k = 0; k < nnb; ++k) {
            if (!(s_nb[k] > 0.0)) continue;
            int rank_hit = 1;
            for (int q = 0; q < nnb; ++q)
                if (q != k && s_nb[q] > 0.0 && (s_nb[q] > s_nb[k] || (s_nb[q] == s_nb[k] && t_nb[q] < t_nb[k]))) ++rank_hit;
            sum += (double)rank_hit / (double)(1 + cnt_nb[k]);
            ++H;
        }
        ap_out[blockIdx.x] = H ? sum / H : 0.0;
    }
}

}  // namespace

extern "C" int gemhip_eval_sampled_ap(int64_t n, int32_t da, int32_t ld, const float *A_host, const float *B_host, const int64_t *row_ptr,
                                      const int32_t *col, int32_t undirected, int64_t nsample, const int32_t *nodes, double *ap_out)
{
    GEMHIP_REQUIRE(n >= 1 && da >= 1 && da <= 512 && ld >= da && A_host && row_ptr && nsample >= 0 && (nsample == 0 || (nodes && ap_out)),
                   "eval_sampled_ap: bad arguments (da <= 512)");
    if (nsample == 0) return GEMHIP_OK;
    for (int64_t k = 0; k < nsample; ++k) GEMHIP_REQUIRE(nodes[k] >= 0 && nodes[k] < n, "eval_sampled_ap: node %d outside [0,%lld)", nodes[k], (long long)n);
    const int64_t nnz = row_ptr[n];
    float *dA = nullptr, *dB = nullptr; int64_t *drp = nullptr; int32_t *dcol = nullptr, *dnodes = nullptr; double *dap = nullptr; int *derr = nullptr;
    int rc = GEMHIP_OK;
    auto cleanup = [&]() { hipFree(dA); if (dB != dA) hipFree(dB); hipFree(drp); hipFree(dcol); hipFree(dnodes); hipFree(dap); hipFree(derr); };
#define EV_TRY(x) do { hipError_t _e = (x); if (_e != hipSuccess) { cleanup(); return fail(GEMHIP_E_HIP, "eval_sampled_ap: %s: %s", #x, hipGetErrorString(_e)); } } while (0)
    const size_t mat = (size_t)n * ld * sizeof(float);
    EV_TRY(hipMalloc((void **)&dA, mat)); EV_TRY(hipMemcpy(dA, A_host, mat, hipMemcpyHostToDevice));
    if (B_host && B_host != A_host) { EV_TRY(hipMalloc((void **)&dB, mat)); EV_TRY(hipMemcpy(dB, B_host, mat, hipMemcpyHostToDevice)); } else dB = dA;
    EV_TRY(hipMalloc((void **)&drp, (n + 1) * 8)); EV_TRY(hipMemcpy(drp, row_ptr, (n + 1) * 8, hipMemcpyHostToDevice));
    EV_TRY(hipMalloc((void **)&dcol, std::max<int64_t>(nnz, 1) * 4)); if (nnz) EV_TRY(hipMemcpy(dcol, col, nnz * 4, hipMemcpyHostToDevice));
    EV_TRY(hipMalloc((void **)&dnodes, nsample * 4)); EV_TRY(hipMemcpy(dnodes, nodes, nsample * 4, hipMemcpyHostToDevice));
    EV_TRY(hipMalloc((void **)&dap, nsample * 8)); EV_TRY(hipMalloc((void **)&derr, 4)); EV_TRY(hipMemset(derr, 0, 4));
    const int nv = (da + WAVE - 1) / WAVE;
#define EV_LAUNCH(NV) hipLaunchKernelGGL((eval_ap_kernel<NV>), dim3((unsigned)nsample), dim3(EV_BLOCK), 0, 0, n, (int)da, dA, dB, (int)ld, drp, dcol, \
                                         (int)undirected, dnodes, dap, derr)
    if (nv <= 1) EV_LAUNCH(1); else if (nv <= 2) EV_LAUNCH(2); else if (nv <= 4) EV_LAUNCH(4); else EV_LAUNCH(8);
#undef EV_LAUNCH
    EV_TRY(hipGetLastError());
    int herr = 0;
    EV_TRY(hipMemcpy(&herr, derr, 4, hipMemcpyDeviceToHost));
    EV_TRY(hipMemcpy(ap_out, dap, nsample * 8, hipMemcpyDeviceToHost));
#undef EV_TRY
    cleanup();
    if (herr) rc = fail(GEMHIP_E_UNSUPPORTED, "eval_sampled_ap: a sampled node has more than %d candidate neighbours; sample other nodes", EV_MAXNB);
    return rc;
}
