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
// One workgroup per (sampled node, chunk of <= 512 of its true neighbours) streams all rows B_j once (a wavefront per row,
// lanes across the d columns, fp64 dot of the fp32 inputs so ranks agree with the float64 reference), and every lane
// compares the score of the row with "its" neighbours' scores -- O(n d) per chunk, exact, no n x n matrix.  The kernel
// returns (score, rank_all - 1) per true neighbour; rank_hit is a sort of <= deg(i) numbers, done on the host.
// score(i, j) = A_i . B_j : A = B = X for GF / node2vec (X_i . X_j), A = X[:, :k], B = X[:, k:] for HOPE (hope.py:43-44).
#include "common.hpp"
#include <algorithm>
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
                                                           int undirected, const int32_t *__restrict__ chunk_node, const int64_t *__restrict__ chunk_off,
                                                           const int32_t *__restrict__ chunk_cnt, const int32_t *__restrict__ nb,
                                                           double *__restrict__ s_out, int32_t *__restrict__ cnt_out)
{
    // One workgroup per (sampled node, chunk of <= EV_MAXNB of its true neighbours): hubs of a power-law graph take several chunks.
    __shared__ double s_nb[EV_MAXNB];
    __shared__ int t_nb[EV_MAXNB];
    __shared__ int cnt_nb[EV_MAXNB];
    const int lane = lane_id(), wave = threadIdx.x >> 6;
    const int i = chunk_node[blockIdx.x];
    const int64_t off = chunk_off[blockIdx.x];
    const int nnb = chunk_cnt[blockIdx.x];
    for (int k = threadIdx.x; k < EV_MAXNB; k += EV_BLOCK) { cnt_nb[k] = 0; t_nb[k] = k < nnb ? nb[off + k] : -1; }
    __syncthreads();
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
    // ---- stream the candidates (j != i, and j > i when undirected: evaluation_util.py:28-35)
    const int64_t lo = undirected ? (int64_t)i + 1 : 0;
    const int nreg_used = (nnb + WAVE - 1) / WAVE;
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
    __syncthreads();
    for (int k = threadIdx.x; k < nnb; k += EV_BLOCK) { s_out[off + k] = s_nb[k]; cnt_out[off + k] = cnt_nb[k]; }
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
    GEMHIP_REQUIRE(nnz == 0 || col, "eval_sampled_ap: col is null");
    // ---- candidate true neighbours of every sampled node (j != i, j > i when undirected, each once), cut into chunks
    std::vector<int32_t> nb, chunk_node, chunk_cnt; std::vector<int64_t> chunk_off, node_off(nsample + 1, 0);
    std::vector<int32_t> tmp;
    for (int64_t k = 0; k < nsample; ++k) {
        const int i = nodes[k];
        tmp.clear();
        for (int64_t e = row_ptr[i]; e < row_ptr[i + 1]; ++e) {
            const int t = col[e];
            GEMHIP_REQUIRE(t >= 0 && t < n, "eval_sampled_ap: column %d outside [0,%lld)", t, (long long)n);
            if (t == i || (undirected && t < i)) continue;
            tmp.push_back(t);
        }
        std::sort(tmp.begin(), tmp.end());
        tmp.erase(std::unique(tmp.begin(), tmp.end()), tmp.end());
        for (size_t o = 0; o < tmp.size(); o += EV_MAXNB) {
            chunk_node.push_back(i); chunk_off.push_back((int64_t)nb.size() + (int64_t)o);
            chunk_cnt.push_back((int32_t)std::min<size_t>(EV_MAXNB, tmp.size() - o));
        }
        nb.insert(nb.end(), tmp.begin(), tmp.end());
        node_off[k + 1] = (int64_t)nb.size();
    }
    const int64_t nchunk = (int64_t)chunk_node.size(), total = (int64_t)nb.size();
    std::vector<double> s_host(std::max<int64_t>(total, 1)); std::vector<int32_t> cnt_host(std::max<int64_t>(total, 1));
    if (nchunk > 0) {
        float *dA = nullptr, *dB = nullptr; int32_t *dnb = nullptr, *dcn = nullptr, *dcc = nullptr, *dcnt = nullptr; int64_t *dco = nullptr; double *ds = nullptr;
        auto cleanup = [&]() { hipFree(dA); if (dB != dA) hipFree(dB); hipFree(dnb); hipFree(dcn); hipFree(dcc); hipFree(dco); hipFree(dcnt); hipFree(ds); };
#define EV_TRY(x) do { hipError_t _e = (x); if (_e != hipSuccess) { cleanup(); return fail(GEMHIP_E_HIP, "eval_sampled_ap: %s: %s", #x, hipGetErrorString(_e)); } } while (0)
        const size_t mat = (size_t)n * ld * sizeof(float);
        EV_TRY(hipMalloc((void **)&dA, mat)); EV_TRY(hipMemcpy(dA, A_host, mat, hipMemcpyHostToDevice));
        if (B_host && B_host != A_host) { EV_TRY(hipMalloc((void **)&dB, mat)); EV_TRY(hipMemcpy(dB, B_host, mat, hipMemcpyHostToDevice)); } else dB = dA;
        EV_TRY(hipMalloc((void **)&dnb, total * 4)); EV_TRY(hipMemcpy(dnb, nb.data(), total * 4, hipMemcpyHostToDevice));
        EV_TRY(hipMalloc((void **)&dcn, nchunk * 4)); EV_TRY(hipMemcpy(dcn, chunk_node.data(), nchunk * 4, hipMemcpyHostToDevice));
        EV_TRY(hipMalloc((void **)&dcc, nchunk * 4)); EV_TRY(hipMemcpy(dcc, chunk_cnt.data(), nchunk * 4, hipMemcpyHostToDevice));
        EV_TRY(hipMalloc((void **)&dco, nchunk * 8)); EV_TRY(hipMemcpy(dco, chunk_off.data(), nchunk * 8, hipMemcpyHostToDevice));
        EV_TRY(hipMalloc((void **)&ds, total * 8)); EV_TRY(hipMalloc((void **)&dcnt, total * 4));
        const int nv = (da + WAVE - 1) / WAVE;
#define EV_LAUNCH(NV) hipLaunchKernelGGL((eval_ap_kernel<NV>), dim3((unsigned)nchunk), dim3(EV_BLOCK), 0, 0, n, (int)da, dA, dB, (int)ld, (int)undirected, \
                                         dcn, dco, dcc, dnb, ds, dcnt)
        if (nv <= 1) EV_LAUNCH(1); else if (nv <= 2) EV_LAUNCH(2); else if (nv <= 4) EV_LAUNCH(4); else EV_LAUNCH(8);
#undef EV_LAUNCH
        EV_TRY(hipGetLastError());
        EV_TRY(hipMemcpy(s_host.data(), ds, total * 8, hipMemcpyDeviceToHost));
        EV_TRY(hipMemcpy(cnt_host.data(), dcnt, total * 4, hipMemcpyDeviceToHost));
#undef EV_TRY
        cleanup();
    }
    // ---- AP_i = mean over true neighbours with s > 0 of rank_hit / rank_all ; rank_hit = position among the neighbours in
    //      the evaluator's order (score descending, node id ascending on ties), rank_all = 1 + the streamed count
    std::vector<int64_t> ord;
    for (int64_t k = 0; k < nsample; ++k) {
        ord.clear();
        for (int64_t e = node_off[k]; e < node_off[k + 1]; ++e) if (s_host[e] > 0.0) ord.push_back(e);
        std::sort(ord.begin(), ord.end(), [&](int64_t x, int64_t y) { return s_host[x] > s_host[y] || (s_host[x] == s_host[y] && nb[x] < nb[y]); });
        double sum = 0.0;
        for (size_t r = 0; r < ord.size(); ++r) sum += (double)(r + 1) / (double)(1 + cnt_host[ord[r]]);
        ap_out[k] = ord.empty() ? 0.0 : sum / (double)ord.size();
    }
    return GEMHIP_OK;
}
