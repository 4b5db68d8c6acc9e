// n2v.hip -- node2vec for MI355X (gfx950): transition tables, biased walks, vocabulary
// counts, and skip-gram with negative sampling (SGNS).
//
// Replaces gem/embedding/node2vec.py:27-54, i.e. the subprocess call of the SNAP binary
// gem/c_exe/node2vec (`-i -o -d -l -r -k -e -p -q -v -dr -w`) and the text files around
// it.  Phases (symbols of the ELF) and their device equivalents:
//   PreprocessTransitionProbs/GetNodeAlias -> n2v_alias_rows_kernel (first-order Vose tables;
//        2nd-order bias by rejection sampling inside the walk, no sum(deg^2) tables)
//   SimulateWalk/AliasDrawInt             -> n2v_walk_kernel   (one lane = one walker)
//   LearnVocab                            -> n2v_vocab_kernel
//   InitUnigramTable                      -> host Vose in fp64 (O(n), once)
//   InitPosEmb/InitNegEmb                 -> sgns_init_kernel
//   TrainModel (Hogwild over walks)       -> sgns_win_kernel / sgns_kernel (sgns.hpp; one wavefront = one walk), instantiated in
//        sgns_hogwild.hip, sgns_det.hip and sgns_part.hip (one bucket of the partitioned N-GPU schedule); this file holds the C ABI and the launch rule
//
// All randomness is a counter-based Philox4x32-10 stream keyed by (seed, walk, position,
// purpose): results do not depend on scheduling, and the CPU oracle reproduces walks,
// counts and alias tables bit for bit.
#include "sgns.hpp"
#include <hipcub/hipcub.hpp>
#include <vector>
#include <cstring>
#include <cmath>
#include <algorithm>

using namespace gemhip;

namespace {
// Stateless stand-in for SNAP's NIdsV.Shuffle(): 4-round Feistel bijection, cycle-walked into [0,n).
__host__ __device__ __forceinline__ uint32_t perm_node(uint32_t j, uint32_t n, uint32_t hb, uint64_t key)
{
    const uint32_t mask = (1u << hb) - 1u;
    do {
        uint32_t L = j >> hb, R = j & mask;
#pragma unroll
        for (uint32_t r = 0; r < 4; ++r) {
            const uint32_t k = (uint32_t)(key >> ((r & 1u) * 32)) + r * 0x9E3779B9u;
            const uint32_t t = L ^ (fmix32(R + k) & mask);
            L = R; R = t;
        }
        j = (L << hb) | R;
    } while (j >= n);
    return j;
}
}  // namespace

// How concentrated the row traffic of TrainModel is: what the Hogwild launch rule (plan_sgns_launch) needs to know about the vocabulary.
struct VocabStats {
    double n = 0.0, total = 0.0, max = 0.0, active = 0.0;     // table rows, tokens, largest count, nodes that occur at all
    double z = 0.0;                    // sum of count^0.75
    double n_eff = 0.0;                // 1 / sum_v q_v^2, q = unigram^0.75 distribution: the table size a uniform graph with the same collision rate would have
    double touch2 = 0.0;               // sum_v (p_v + 5 q_v)^2, p = token share: collision rate of the rows a (centre, context) pair TOUCHES (its context row + five negatives)
    mutable std::vector<int32_t> cnt_desc;     // token counts, descending (after sort_once), and ...
    mutable std::vector<double> u2_prefix;     // ... u2_prefix[i] = sum of (count^0.75)^2 over the i largest counts
    mutable std::vector<double> c_prefix, u_prefix;   // ... sums of count and of count^0.75 over the i largest counts
    void build(const int32_t *cnt, int64_t len)
    {
        double tot = 0.0, mx = 0.0, zz = 0.0, z2 = 0.0, c2 = 0.0, cu = 0.0;
        n = (double)len; active = 0.0;
        for (int64_t i = 0; i < len; ++i) {
            const int32_t c = cnt[i];
            if (c <= 0) continue;
            tot += c; mx = std::max(mx, (double)c); active += 1.0;
            const double u = std::pow((double)c, 0.75);
            zz += u; z2 += u * u; c2 += (double)c * c; cu += (double)c * u;
        }
        total = tot; max = mx; z = zz;
        n_eff = z2 > 0.0 ? zz * zz / z2 : n;
        touch2 = (tot > 0.0 && zz > 0.0) ? c2 / (tot * tot) + 10.0 * cu / (tot * zz) + 25.0 * z2 / (zz * zz) : 0.0;
        cnt_desc.assign(cnt, cnt + len);                     // sorted (and the prefix sums formed) only if the rule has to look at the cold rows
        u2_prefix.clear();
    }
    void sort_once() const
    {
        if (!u2_prefix.empty()) return;
        std::sort(cnt_desc.begin(), cnt_desc.end(), std::greater<int32_t>());
        u2_prefix.assign(cnt_desc.size() + 1, 0.0);
        c_prefix.assign(cnt_desc.size() + 1, 0.0); u_prefix.assign(cnt_desc.size() + 1, 0.0);
        for (size_t i = 0; i < cnt_desc.size(); ++i) {
            const double u = cnt_desc[i] > 0 ? std::pow((double)cnt_desc[i], 0.75) : 0.0;
            u2_prefix[i + 1] = u2_prefix[i] + u * u;
            c_prefix[i + 1] = c_prefix[i] + (cnt_desc[i] > 0 ? (double)cnt_desc[i] : 0.0);
            u_prefix[i + 1] = u_prefix[i] + u;
        }
    }
    // hot row-operations per trained (centre, context) pair when the rows with count >= thr are hot: P(the context is hot) + 5 x P(a negative is hot)
    double hot_ops_per_pair(double thr) const
    {
        if (cnt_desc.empty() || z <= 0.0 || total <= 0.0) return 0.0;
        sort_once();
        const size_t nhot = (size_t)(std::lower_bound(cnt_desc.begin(), cnt_desc.end(), thr, [](int32_t c, double t) { return (double)c >= t; }) - cnt_desc.begin());
        return c_prefix[nhot] / total + 5.0 * u_prefix[nhot] / z;
    }
    // effective table size of the negative-sampling distribution over the rows that are NOT hot (count < thr): hot rows take atomic adds and lose nothing
    double n_eff_cold(double thr) const
    {
        if (cnt_desc.empty() || z <= 0.0) return n;
        sort_once();
        const size_t nhot = (size_t)(std::lower_bound(cnt_desc.begin(), cnt_desc.end(), thr, [](int32_t c, double t) { return (double)c >= t; }) - cnt_desc.begin());
        const double s2 = (u2_prefix.back() - u2_prefix[nhot]) / (z * z);
        return s2 > 0.0 ? 1.0 / s2 : 1e300;
    }
};

// The knobs of a handle that steer the SGNS launch (gemhip_n2v_set_max_waves, gemhip_sgns_set_window_cache / _hogwild / _hot_rows) ...
struct SgnsKnobs {
    int32_t max_waves = 0;            // 0 = auto (plan_sgns_launch)
    int32_t cache_radius = -1;        // sgns_win_kernel LDS window radius: -1 auto, 0 = off (sgns_kernel)
    int32_t cache_delta = -1;         // -1 auto, 0 overwrite on leave, 1 delta write-back
    int32_t prefetch = 2;             // pairs whose negative rows are requested ahead: 2 (default) or 1 (d == 64/128/256.., whole window cached)
    int32_t reload = 1;               // Hogwild launches: update negative rows as they are at store time (second fetch) and the centre row by atomic add
    int32_t hot_count = -1;           // nodes with at least this many tokens never enter the LDS window: -1 auto (tokens / ((W-1) x (2R+1))), 0 off
    int32_t window_span = 0;          // positions of a walk whose context rows a wavefront holds at once: 0 = 2R+1 (the sliding window); the whole walk in the bucket kernel's whole-walk mode
    bool part = false;                // a bucket launch of the partitioned schedule (sgns_win_kernel<PART>: as-loaded window copies in global scratch, 3 wavefronts per SIMD)
    double duty = 1.0;                // fraction of a wavefront's time spent in pair steps (negative rows open); < 1 only for the buckets of the partitioned schedule
    double touch_scale = 1.0;         // factor on VocabStats::touch2 (bucket launches: the pairs of ONE bucket touch the rows of two partitions only: parts x duty)
    bool has_local_hot = false;       // this corpus HAS locally hot nodes (ensure_hotkey): the launch carries the staging row and a hot threshold even without count-hot rows
    int32_t local_hot = 8;            // LOCALLY HOT ROWS: a node with at least this many tokens per walk that contains it is treated as a hot row (0: off)
    int32_t neg_count = 0;            // fresh bit 2: token count from which a negative row is updated by atomic add (0: every row)
    int32_t fresh = 0;                // FRESH HOT ROWS (sgns.hpp, SgnsArgs::fresh): bit 0 hot centre words by returning atomics, bit 1 hot negatives re-read before the dot products
    bool node_id_layout = false;      // the unigram table the launch draws from is in node-id order (gemhip_n2v_build_unigram), not the binary's: half the concurrent-touch bound (plan_sgns_launch)
};
// ... and what a launch of TrainModel over `nwalks` walks then looks like (pure host arithmetic: gemhip_sgns_plan_launch exposes it to the CPU tests)
struct SgnsLaunchPlan {
    bool window = false;              // sgns_win_kernel (LDS window) or sgns_kernel
    bool delta = false;               // window kernel: delta write-back (the Hogwild instantiation)
    int R = 0;                        // window radius
    int64_t waves = 1;                // concurrent wavefronts = concurrent walks
    int32_t hot_thr = 0;              // token count from which a row is "hot" (0: none)
    size_t lds = 0;                   // dynamic LDS bytes per workgroup
    int blocks = 1, threads = 64;
};

struct gemhip_n2v {
    int64_t n = 0, nnz = 0;
    int device = 0;
    bool uniform_rows = true;         // every row has equal weights -> no alias tables needed
    int64_t *d_row_ptr = nullptr;
    int32_t *d_col = nullptr;         // columns sorted inside each row
    // walk start nodes: the nodes that occur in the edge list (the reference binary only knows those; an isolated node
    // never reaches it).  start[0..m_start) ascending; walk id r*m_start + j starts at start[perm_r(j)]
    int64_t m_start = 0;
    int32_t *d_start = nullptr;
    float *d_w = nullptr;
    std::vector<int32_t> hub_rows;    // rows with at least ALIAS_HUB_DEG neighbours (n2v_alias_hub_kernel builds their tables: a workgroup per row)
    std::vector<int64_t> hub_off;     // ... and the offset of each one's scratch segment (prefix sum of their degrees)
    float *d_U = nullptr;             // first-order alias tables (per-row segments)
    int32_t *d_K = nullptr;
    // walks
    int32_t *d_walks = nullptr;
    int64_t walks_cap = 0;            // tokens allocated
    int64_t nwalks = 0;               // local walks held
    int32_t walk_len = 0;
    int64_t walk_id_offset = 0;       // global id of local walk 0
    // vocabulary / unigram
    int32_t *d_counts = nullptr;
    float *d_UT = nullptr;
    int32_t *d_KT = nullptr;
    uint2 *d_UK = nullptr;             // {bits of UT[i], KT[i]} interleaved: one 8-byte gather instead of two 4-byte gathers
    bool unigram_ready = false;
    // slot tables of the window kernels (SgnsArgs::SK): {X, UT[X], KT[X]} by SLOT, built on the device from the tables above on the first launch
    // after a table build, once per RndUnigramInt-quirk setting (sk_state / skp_state: -1 stale, else the quirk bit they were built for)
    uint4 *d_SK = nullptr; int64_t sk_cap = 0; int sk_state = -1;
    uint4 *d_SKp = nullptr; int skp_state = -1;
    // vocabulary-order layout (gemhip_n2v_build_unigram_vocab_order): the slot table of RndUnigramInt, and how many slots it has
    int32_t *d_KTslot = nullptr; int64_t n_vocab = 0; bool vocab_order = false;
    unsigned long long *d_first = nullptr;   // first token index of every node (scratch of that builder)
    // embeddings
    int32_t d = 0;
    float *SynPos = nullptr, *SynNeg = nullptr;
    bool own_syn = false;
    SgnsKnobs kn;                     // launch knobs (setters below; environment overrides read once in gemhip_n2v_create)
    int32_t *d_hotkey = nullptr; int32_t *d_wcount = nullptr; unsigned int *d_nlocal = nullptr; int hotkey_state = 0; int64_t n_local_hot = 0;   // LOCALLY HOT ROWS (ensure_hotkey)
    SgnsLaunchPlan last_plan;         // what the last gemhip_sgns_train / _train_part on this handle actually launched (gemhip_sgns_last_launch)
    int32_t last_fresh = 0;
    VocabStats vs;                    // vocabulary statistics (gemhip_n2v_build_unigram*): how concentrated the row traffic is -> plan_sgns_launch
    float *d_dummy = nullptr; size_t dummy_bytes = 0;   // sgns_win_kernel: one scratch row per wavefront
    float *d_scratch = nullptr; size_t scratch_bytes = 0;   // sgns_win_kernel<PART>: the window rows as loaded, 2R+1 rows per wavefront
    unsigned long long *d_pairs = nullptr;   // (centre,context) pairs trained so far
    // per-partition unigram tables (multi-GPU episode schedule): partition p = {v : v % parts == p}, local index v / parts
    int32_t parts = 0;
    float *d_UTp = nullptr; int32_t *d_KTp = nullptr;
    uint2 *d_UKp = nullptr;                  // {bits of UTp[i], KTp[i]} interleaved (sgns_win_kernel<PART>)
    // ... in the binary's layout (gemhip_n2v_build_unigram_parts_vocab_order): per partition the slot table of RndUnigramInt in LOCAL indices
    // (d_KTslotp[part_off[p] + slot]) and its number of slots (the partition's nodes that occur); the alias arrays above are then indexed by local row
    int32_t *d_KTslotp = nullptr; std::vector<int64_t> part_slots; bool parts_vocab_order = false;
    std::vector<int64_t> part_off;           // table p occupies [part_off[p], part_off[p+1])
    std::vector<VocabStats> vs_part;         // vocabulary statistics of each partition's rows (launch rule of gemhip_sgns_train_part)
    bool own_counts = true;
};

namespace {

// ------------------------------------------------------------------ alias tables
constexpr int ALIAS_HUB_DEG = 2048;      // rows from this many neighbours on take n2v_alias_hub_kernel (oracle: ORACLE_ALIAS_HUB_DEG)
// GetNodeAlias (ELF @0x4115f0): one thread per CSR row, sequential Vose in fp32.  The two
// stacks share the row's segment of `work` (small grows up from 0, large down from N-1).
__global__ void n2v_alias_rows_kernel(int64_t n, const int64_t *__restrict__ row_ptr, const float *__restrict__ w,
                                      float *__restrict__ U, int32_t *__restrict__ K, int32_t *__restrict__ work_all)
{
    const int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= n) return;
    const int64_t a = row_ptr[v];
    const int32_t N = (int32_t)(row_ptr[v + 1] - a);
    if (N == 0 || N >= ALIAS_HUB_DEG) return;                  // (hub rows: n2v_alias_hub_kernel)
    const float *wt = w + a; float *Ur = U + a; int32_t *Kr = K + a; int32_t *work = work_all + a;
    float sum = 0.0f;
    for (int32_t i = 0; i < N; ++i) sum += wt[i];
    int32_t ns = 0, nl = 0;
    for (int32_t i = 0; i < N; ++i) {
        Kr[i] = 0;
        const float u = (wt[i] / sum) * (float)N;
        Ur[i] = u;
        if (u < 1.0f) work[ns++] = i; else work[N - 1 - nl++] = i;
    }
    while (ns > 0 && nl > 0) {
        const int32_t s = work[--ns];
        const int32_t l = work[N - nl]; --nl;
        Kr[s] = l;
        const float u = Ur[l] + Ur[s] - 1.0f;
        Ur[l] = u;
        if (u < 1.0f) work[ns++] = l; else work[N - 1 - nl++] = l;
    }
    while (ns > 0) Ur[work[--ns]] = 1.0f;
    while (nl > 0) { Ur[work[N - nl]] = 1.0f; --nl; }
}

// GetNodeAlias for HUB rows (N >= ALIAS_HUB_DEG): one workgroup per row, no N-step chain.  The stacks of GetNodeAlias are filled in index order and
// popped from the back, so smalls and larges are met in descending index order and the loop's outcome is closed form in two prefix sums -- D_j, the
// deficits of the first j smalls, and E_m, the excesses of the first m larges (both in stack order):
//   small s_j -> K = L_m, m = min{m : E_m >= D_{j-1}};   large L_m turns small at the first j with D_j > E_m: U = 1 + E_m - D_j, K = L_{m+1}
// (derivation and the proof by test against the sequential loop: oracle/n2v_oracle.c oracle_alias_build_hub, tests/test_oracle_n2v.py).  fp64, with
// the order of every sum FIXED exactly as the oracle fixes it -- chunks of 256 entries in stack order, sequential inside a chunk (one thread per
// chunk), chunk totals accumulated sequentially (thread 0), entry = chunk base + running sum -- so the table is the oracle's bit for bit whatever the
// block size.  scr: per row 2 N int32 (S, L) then 2 N double (D, E) then per-chunk {double d, e; int32 s, l}; a 94 115-neighbour R-MAT hub takes
// ~0.1 ms instead of a 94k-step dependent chain on one lane.
constexpr int ALIAS_CHUNK = 256;
__global__ __launch_bounds__(256) void n2v_alias_hub_kernel(const int32_t *__restrict__ hubs, const int64_t *__restrict__ hub_off, const int64_t *__restrict__ row_ptr,
                                                            const float *__restrict__ w, float *__restrict__ U, int32_t *__restrict__ K, unsigned char *__restrict__ scr_all)
{
#pragma clang fp contract(off)          // the oracle is built with -ffp-contract=off: `1.0 - (w / total) * N` must round the product first, here too
    const int32_t v = hubs[blockIdx.x];
    const int64_t a = row_ptr[v];
    const int32_t N = (int32_t)(row_ptr[v + 1] - a);
    const int32_t nch = (N + ALIAS_CHUNK - 1) / ALIAS_CHUNK;
    const float *wt = w + a; float *Ur = U + a; int32_t *Kr = K + a;
    // scratch segment of this row: hub_off counts entries; every entry owns 24 bytes, every chunk 24 more (chunks of row b start after all entries)
    unsigned char *scr = scr_all + (size_t)hub_off[blockIdx.x] * 24 + (size_t)(hub_off[blockIdx.x] / ALIAS_CHUNK + blockIdx.x) * 24;
    double *D = reinterpret_cast<double *>(scr), *E = D + N;
    int32_t *S = reinterpret_cast<int32_t *>(E + N), *L = S + N;
    double *cd = reinterpret_cast<double *>(scr + (size_t)N * 24), *ce = cd + nch;
    int32_t *cs = reinterpret_cast<int32_t *>(ce + nch), *cl = cs + nch;
    __shared__ double sh_total;
    __shared__ int32_t sh_ns, sh_nl;
    // total weight: chunk sums in INDEX order (into cd), accumulated sequentially
    for (int32_t c = threadIdx.x; c < nch; c += blockDim.x) {
        double t = 0.0;
        const int32_t i1 = (c + 1) * ALIAS_CHUNK < N ? (c + 1) * ALIAS_CHUNK : N;
        for (int32_t i = c * ALIAS_CHUNK; i < i1; ++i) t += (double)wt[i];
        cd[c] = t;
    }
    __syncthreads();
    if (threadIdx.x == 0) { double t = 0.0; for (int32_t c = 0; c < nch; ++c) t += cd[c]; sh_total = t; }
    __syncthreads();
    const double total = sh_total, dN = (double)N;
    // per chunk of the STACK order (r = 0 is the top of both stacks, i = N - 1 - r): deficit / excess sums and counts
    for (int32_t c = threadIdx.x; c < nch; c += blockDim.x) {
        double ds = 0.0, es = 0.0; int32_t ns = 0, nl = 0;
        const int32_t r1 = (c + 1) * ALIAS_CHUNK < N ? (c + 1) * ALIAS_CHUNK : N;
        for (int32_t r = c * ALIAS_CHUNK; r < r1; ++r) {
            const double u = ((double)wt[N - 1 - r] / total) * dN;
            if (u < 1.0) { ds += 1.0 - u; ++ns; } else { es += u - 1.0; ++nl; }
        }
        cd[c] = ds; ce[c] = es; cs[c] = ns; cl[c] = nl;
    }
    __syncthreads();
    if (threadIdx.x == 0) {                                  // exclusive prefix over the chunks, sequential
        double bd = 0.0, be = 0.0; int32_t bs = 0, bl = 0;
        for (int32_t c = 0; c < nch; ++c) {
            const double ds = cd[c], es = ce[c]; const int32_t ns = cs[c], nl = cl[c];
            cd[c] = bd; ce[c] = be; cs[c] = bs; cl[c] = bl;
            bd += ds; be += es; bs += ns; bl += nl;
        }
        sh_ns = bs; sh_nl = bl;
    }
    __syncthreads();
    for (int32_t c = threadIdx.x; c < nch; c += blockDim.x) {
        double ds = 0.0, es = 0.0; int32_t ns = cs[c], nl = cl[c];
        const double bd = cd[c], be = ce[c];
        const int32_t r1 = (c + 1) * ALIAS_CHUNK < N ? (c + 1) * ALIAS_CHUNK : N;
        for (int32_t r = c * ALIAS_CHUNK; r < r1; ++r) {
            const int32_t i = N - 1 - r;
            const double u = ((double)wt[i] / total) * dN;
            Kr[i] = 0;
            if (u < 1.0) { ds += 1.0 - u; S[ns] = i; D[ns] = bd + ds; ++ns; Ur[i] = (float)u; }
            else { es += u - 1.0; L[nl] = i; E[nl] = be + es; ++nl; Ur[i] = 1.0f; }
        }
    }
    __threadfence_block();
    __syncthreads();
    const int32_t NS = sh_ns, NL = sh_nl;
    for (int32_t j = threadIdx.x; j < NS; j += blockDim.x) {  // smalls: first m with E[m] >= D[j-1]
        const double dp = j ? D[j - 1] : 0.0;
        int32_t lo = 0, hi = NL;
        while (lo < hi) { const int32_t mid = (lo + hi) >> 1; if (E[mid] >= dp) hi = mid; else lo = mid + 1; }
        if (lo < NL) Kr[S[j]] = L[lo]; else Ur[S[j]] = 1.0f;
    }
    for (int32_t m = threadIdx.x; m < NL; m += blockDim.x) {  // larges: first j with D[j] > E[m]
        const double em = E[m];
        int32_t lo = 0, hi = NS;
        while (lo < hi) { const int32_t mid = (lo + hi) >> 1; if (D[mid] > em) hi = mid; else lo = mid + 1; }
        if (lo < NS && m + 1 < NL) { Ur[L[m]] = (float)(1.0 + em - D[lo]); Kr[L[m]] = L[m + 1]; }
    }
}

// ------------------------------------------------------------------------- walks
__device__ __forceinline__ bool has_edge_sorted(const int64_t *__restrict__ row_ptr, const int32_t *__restrict__ col, int32_t t, int32_t x)
{
    int64_t lo = row_ptr[t];
    const int64_t end = row_ptr[t + 1];
    int64_t hi = end;
    while (lo < hi) {
        const int64_t mid = (lo + hi) >> 1;
        if (col[mid] < x) lo = mid + 1; else hi = mid;
    }
    return lo < end && col[lo] == x;
}

// SimulateWalk (ELF @0x411a00).  One lane per walk; tokens are buffered 16 at a time in
// registers so each lane writes 64 contiguous bytes (dwordx4 stores) instead of 4-byte scatters.
template <bool SECOND, bool WEIGHTED>
__global__ __launch_bounds__(256) void n2v_walk_kernel(int64_t n, int64_t m, const int32_t *__restrict__ start, uint32_t hb, const int64_t *__restrict__ row_ptr,
                                                       const int32_t *__restrict__ col, const float *__restrict__ U,
                                                       const int32_t *__restrict__ K, float ip, float iq, float amax,
                                                       int32_t walk_len, uint64_t seed, int32_t flags, int64_t walk_begin,
                                                       int64_t count, int32_t *__restrict__ walks)
{
    const int64_t tid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (tid >= count) return;
    const int64_t wid = walk_begin + tid;
    const uint32_t round = (uint32_t)(wid / m), j = (uint32_t)(wid % m);
    int32_t cur = start[perm_node(j, (uint32_t)m, hb, seed ^ ((uint64_t)(round + 1) * 0x9E3779B97F4A7C15ull))];
    int32_t prev = -1;
    const int32_t pad = (flags & 1) ? 0 : -1;
    const bool uniform_first = (flags & 8) != 0;
    bool alive = true;
    int32_t *out = walks + tid * walk_len;
    const bool vec_ok = (walk_len & 3) == 0;

    for (int32_t base = 0; base < walk_len; base += 16) {
        int32_t buf[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const int32_t len = base + k;       // number of tokens already emitted == index of this token
            int32_t tok = pad;
            if (len == 0) tok = cur;
            else if (len < walk_len && alive) {
                const int64_t a = row_ptr[cur];
                const uint32_t deg = (uint32_t)(row_ptr[cur + 1] - a);
                if (deg == 0) alive = false;
                else {
                    int32_t nxt = -1;
                    for (uint32_t trial = 0;; ++trial) {
                        const u32x4 r = philox4x32_10(seed, (uint32_t)wid, (uint32_t)((uint64_t)wid >> 32), (uint32_t)len,
                                                      (uint32_t)TAG_WALK | (trial << 8));
                        uint32_t slot = mulhi_range(r.x, deg);
                        if (WEIGHTED && !(len == 1 && uniform_first)) {
                            if (!(u01(r.y) < U[a + slot])) slot = (uint32_t)K[a + slot];
                        }
                        const int32_t x = col[a + slot];
                        if (!SECOND || len == 1) { nxt = x; break; }
                        const float alpha = (x == prev) ? ip : (has_edge_sorted(row_ptr, col, prev, x) ? 1.0f : iq);
                        if (u01(r.z) * amax < alpha || trial >= 4095u) { nxt = x; break; }
                    }
                    prev = cur; cur = nxt; tok = nxt;
                }
            }
            buf[k] = tok;
        }
        if (vec_ok) {
#pragma unroll
            for (int k = 0; k < 16; k += 4)
                if (base + k < walk_len) *reinterpret_cast<int4 *>(out + base + k) = make_int4(buf[k], buf[k + 1], buf[k + 2], buf[k + 3]);
        } else {
#pragma unroll
            for (int k = 0; k < 16; ++k)
                if (base + k < walk_len) out[base + k] = buf[k];
        }
    }
}

// LearnVocab (ELF @0x40d560): token histogram.
__global__ void n2v_vocab_kernel(const int32_t *__restrict__ walks, int64_t ntokens, int32_t *__restrict__ counts)
{
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < ntokens; i += stride) {
        const int32_t t = walks[i];
        if (t >= 0) atomicAdd(&counts[t], 1);
    }
}

// LOCALLY HOT ROWS (round 6).  wcount[v] = number of walks that CONTAIN v (one wavefront per walk: a token counts at its first occurrence in the walk).
// A node whose tokens are packed into few walks -- count[v] / wcount[v] occurrences per walk that touches it -- sits in that walk's LDS window for the
// whole walk, not for 2R+1 centres: the two nodes of an isolated edge make up ALL 80 tokens of each of their 20 walks.  Two wavefronts that train two of
// those walks at the same time each apply a whole walk's worth of updates (~800 pair steps per row) to the same base and both deltas are added: the pair
// ends with a norm 14 % above its peers' (measured, R-MAT scale 17 at 1 536 wavefronts: profiles/r06_canaries_rmat17.jsonl) and, because the nodes of
// such components are nearly collinear (cosine 0.93; they are only ever pushed away from the same hub rows), outranks the true neighbour of dozens of
// them -- the heavy tail of the power-law "Hogwild bias" (each event costs 2-5 % of the graph's MAP; its probability grows with the width).
__global__ __launch_bounds__(64) void n2v_walk_presence_kernel(const int32_t *__restrict__ walks, int64_t nwalks, int32_t walk_len, int32_t *__restrict__ wcount)
{
    extern __shared__ int32_t tokp[];
    const int lane = threadIdx.x;
    for (int64_t wl = blockIdx.x; wl < nwalks; wl += gridDim.x) {
        const int32_t *walk = walks + wl * walk_len;
        for (int k = lane; k < walk_len; k += WAVE) tokp[k] = walk[k];
        __builtin_amdgcn_wave_barrier();
        for (int k = lane; k < walk_len; k += WAVE) {
            const int32_t v = tokp[k];
            bool first = v >= 0;
            for (int q = 0; q < k && first; ++q) first = tokp[q] != v;
            if (first) atomicAdd(&wcount[v], 1);
        }
        __builtin_amdgcn_wave_barrier();
    }
}

// hotkey[v] = INT32_MAX for a locally hot node (count >= per_walk x walks containing it), else its token count: what the SGNS kernels compare with hot_thr
__global__ void n2v_hotkey_kernel(int64_t n, const int32_t *__restrict__ counts, const int32_t *__restrict__ wcount, int32_t per_walk, int32_t *__restrict__ hotkey,
                                  unsigned int *__restrict__ nlocal)
{
    const int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= n) return;
    const int32_t c = counts[v], w = wcount[v];
    const bool loc = w > 0 && (int64_t)c >= (int64_t)per_walk * w;
    hotkey[v] = loc ? INT32_MAX : c;
    if (loc) atomicAdd(nlocal, 1u);
}

// -------------------------------------------------------------------------- SGNS
// InitPosEmb (ELF @0x40e270): (U(0,1)-0.5)/d ; InitNegEmb: zeros.
// first[v] = index of the first token equal to v (LearnVocab @0x40d560 renames the tokens 0..N-1 in that order)
__global__ void n2v_first_token_kernel(const int32_t *__restrict__ walks, int64_t ntokens, unsigned long long *__restrict__ first)
{
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < ntokens; t += stride) {
        const int32_t v = walks[t];
        if (v >= 0 && first[v] > (unsigned long long)t) atomicMin(&first[v], (unsigned long long)t);
    }
}

__global__ void iota_kernel(int32_t *p, int64_t n)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = (int32_t)i;
}

// SgnsArgs::SK.  RndUnigramInt draws slot = floor(u n), X = KT[slot] (quirk) or slot, then reads {UTable[X], KTable[X]}: the second gather depends on
// the first.  SK[slot] = {X, bits of UTable[X], KTable[X], 0} answers both with one 16-byte read (same values: the draws are unchanged bit for bit).
__global__ void n2v_slot_table_kernel(int64_t nslots, const int32_t *__restrict__ KT, const uint2 *__restrict__ UK, int quirk, uint4 *__restrict__ SK)
{
    const int64_t s = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= nslots) return;
    const int32_t X = quirk ? KT[s] : (int32_t)s;
    const uint2 uk = UK[X];
    SK[s] = make_uint4((uint32_t)X, uk.x, uk.y, 0u);
}

__global__ void sgns_init_kernel(float *SynPos, float *SynNeg, int64_t total, int32_t d, uint64_t seed)
{
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t * 4 >= total) return;
    const u32x4 r = philox4x32_10(seed, (uint32_t)t, (uint32_t)((uint64_t)t >> 32), 0u, (uint32_t)TAG_INIT);
    const uint32_t v[4] = {r.x, r.y, r.z, r.w};
    for (int k = 0; k < 4; ++k)
        if (t * 4 + k < total) {
            SynPos[t * 4 + k] = (u01(v[k]) - 0.5f) / (float)d;
            SynNeg[t * 4 + k] = 0.0f;
        }
}

uint32_t half_bits(uint64_t n)
{
    uint32_t bits = 1;
    while (((uint64_t)1 << bits) < n) ++bits;
    return (bits + 1) / 2;
}

}  // namespace

// ------------------------------------------------------------------- host API
extern "C" int gemhip_n2v_create(int64_t n, int64_t nnz, const int64_t *row_ptr, const int32_t *col, const float *w,
                                 gemhip_n2v_t *out)
{
    GEMHIP_REQUIRE(out != nullptr, "n2v_create: out is NULL");
    *out = nullptr;
    GEMHIP_REQUIRE(n > 0 && n < (int64_t)0x7fffffff, "n2v_create: n=%lld out of range", (long long)n);
    GEMHIP_REQUIRE(nnz >= 0 && row_ptr && (nnz == 0 || col), "n2v_create: bad CSR arrays");
    GEMHIP_REQUIRE(row_ptr[0] == 0 && row_ptr[n] == nnz, "n2v_create: row_ptr[0]=%lld row_ptr[n]=%lld nnz=%lld",
                   (long long)row_ptr[0], (long long)row_ptr[n], (long long)nnz);
    // sort columns inside each row (needed by the has_edge test of the 2nd-order walk); weights follow
    const double t_host0 = phase_now();
    std::vector<int32_t> c(col, col + nnz);
    std::vector<float> ww;
    if (w) ww.assign(w, w + nnz);
    bool uniform = true;
    std::vector<std::pair<int32_t, float>> tmp;
    for (int64_t v = 0; v < n; ++v) {
        const int64_t a = row_ptr[v], b = row_ptr[v + 1];
        GEMHIP_REQUIRE(a <= b && b <= nnz, "n2v_create: row_ptr not monotone at row %lld", (long long)v);
        bool sorted = true;
        for (int64_t e = a; e < b; ++e) {
            GEMHIP_REQUIRE(c[e] >= 0 && c[e] < n, "n2v_create: column %d outside [0,%lld)", c[e], (long long)n);
            if (e > a && c[e] < c[e - 1]) sorted = false;
            if (w && w[e] != w[a]) uniform = false;
            if (w) GEMHIP_REQUIRE(w[e] > 0.f, "n2v_create: non-positive weight at edge %lld", (long long)e);
        }
        if (!sorted) {
            tmp.clear();
            for (int64_t e = a; e < b; ++e) tmp.emplace_back(c[e], w ? ww[e] : 1.f);
            std::stable_sort(tmp.begin(), tmp.end(), [](const auto &x, const auto &y) { return x.first < y.first; });
            for (int64_t e = a; e < b; ++e) { c[e] = tmp[e - a].first; if (w) ww[e] = tmp[e - a].second; }
        }
    }
    std::vector<int32_t> start;
    {
        std::vector<char> present(n, 0);
        for (int64_t v = 0; v < n; ++v)
            if (row_ptr[v + 1] > row_ptr[v]) present[v] = 1;
        for (int64_t e = 0; e < nnz; ++e) present[c[e]] = 1;
        for (int64_t v = 0; v < n; ++v) if (present[v]) start.push_back((int32_t)v);
    }
    auto *h = new gemhip_n2v();
    h->n = n; h->nnz = nnz; h->uniform_rows = uniform; h->m_start = (int64_t)start.size();
    if (!uniform) {
        h->hub_off.push_back(0);
        for (int64_t v = 0; v < n; ++v)
            if (row_ptr[v + 1] - row_ptr[v] >= ALIAS_HUB_DEG) { h->hub_rows.push_back((int32_t)v); h->hub_off.push_back(h->hub_off.back() + (row_ptr[v + 1] - row_ptr[v])); }
    }
    // A/B knobs: read ONCE, here (the launch path reads no environment); the setters below override them per handle
    if (const char *e = getenv("GEMHIP_SGNS_MAX_WAVES")) h->kn.max_waves = std::max(0, atoi(e));
    if (const char *e = getenv("GEMHIP_SGNS_CACHE_R")) h->kn.cache_radius = std::min(31, std::max(-1, atoi(e)));
    if (const char *e = getenv("GEMHIP_SGNS_CACHE_DELTA")) h->kn.cache_delta = std::min(1, std::max(-1, atoi(e)));
    if (const char *e = getenv("GEMHIP_SGNS_PREFETCH")) h->kn.prefetch = std::min(2, std::max(1, atoi(e)));
    if (const char *e = getenv("GEMHIP_SGNS_RELOAD")) h->kn.reload = atoi(e) != 0;
    if (const char *e = getenv("GEMHIP_SGNS_HOT_COUNT")) h->kn.hot_count = std::max(-1, atoi(e));
    if (const char *e = getenv("GEMHIP_SGNS_LOCAL_HOT")) h->kn.local_hot = std::max(0, atoi(e));
    if (const char *e = getenv("GEMHIP_SGNS_FRESH")) h->kn.fresh = atoi(e) & 7;
    if (const char *e = getenv("GEMHIP_SGNS_NEG_COUNT")) h->kn.neg_count = std::max(0, atoi(e));
    if (hipGetDevice(&h->device) != hipSuccess) { delete h; return fail(GEMHIP_E_HIP, "n2v_create: no HIP device"); }
    phase_acc()[PH_HOST] += phase_now() - t_host0;
    PhaseScope ph_up(PH_H2D);
    hipError_t e = hipMalloc((void **)&h->d_row_ptr, (n + 1) * sizeof(int64_t));
    if (e == hipSuccess) e = hipMemcpy(h->d_row_ptr, row_ptr, (n + 1) * sizeof(int64_t), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMalloc((void **)&h->d_col, std::max<int64_t>(nnz, 4) * sizeof(int32_t));
    if (e == hipSuccess && nnz) e = hipMemcpy(h->d_col, c.data(), nnz * sizeof(int32_t), hipMemcpyHostToDevice);
    if (e == hipSuccess && !uniform) {
        e = hipMalloc((void **)&h->d_w, nnz * sizeof(float));
        if (e == hipSuccess) e = hipMemcpy(h->d_w, ww.data(), nnz * sizeof(float), hipMemcpyHostToDevice);
    }
    if (e == hipSuccess) e = hipMalloc((void **)&h->d_start, std::max<size_t>(start.size(), 4) * sizeof(int32_t));
    if (e == hipSuccess && !start.empty()) e = hipMemcpy(h->d_start, start.data(), start.size() * sizeof(int32_t), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMalloc((void **)&h->d_counts, n * sizeof(int32_t));
    if (e == hipSuccess) e = hipMalloc((void **)&h->d_pairs, sizeof(unsigned long long));
    if (e == hipSuccess) e = hipMemset(h->d_pairs, 0, sizeof(unsigned long long));
    if (e != hipSuccess) { gemhip_n2v_destroy(h); return fail(GEMHIP_E_HIP, "n2v_create: device upload failed: %s", hipGetErrorString(e)); }
    *out = h;
    return GEMHIP_OK;
}

extern "C" int gemhip_n2v_destroy(gemhip_n2v_t h)
{
    if (!h) return GEMHIP_OK;
    hipFree(h->d_start);
    hipFree(h->d_row_ptr); hipFree(h->d_col); hipFree(h->d_w); hipFree(h->d_U); hipFree(h->d_K); hipFree(h->d_walks); hipFree(h->d_dummy); hipFree(h->d_scratch);
    if (h->own_counts) hipFree(h->d_counts);
    hipFree(h->d_hotkey); hipFree(h->d_wcount); hipFree(h->d_nlocal);
    hipFree(h->d_UT); hipFree(h->d_KT); hipFree(h->d_UK); hipFree(h->d_SK); hipFree(h->d_SKp); hipFree(h->d_KTslot); hipFree(h->d_first); hipFree(h->d_pairs); hipFree(h->d_UTp); hipFree(h->d_KTp); hipFree(h->d_UKp); hipFree(h->d_KTslotp);
    if (h->own_syn) { hipFree(h->SynPos); hipFree(h->SynNeg); }
    delete h;
    return GEMHIP_OK;
}

extern "C" int gemhip_n2v_build_alias(gemhip_n2v_t h, void *stream)
{
    GEMHIP_REQUIRE(h, "n2v_build_alias: NULL handle");
    if (h->uniform_rows || h->d_U) return GEMHIP_OK;
    int32_t *work = nullptr;
    GEMHIP_CHECK(hipMalloc((void **)&h->d_U, h->nnz * sizeof(float)));
    GEMHIP_CHECK(hipMalloc((void **)&h->d_K, h->nnz * sizeof(int32_t)));
    GEMHIP_CHECK(hipMalloc((void **)&work, h->nnz * sizeof(int32_t)));
    hipLaunchKernelGGL(n2v_alias_rows_kernel, dim3((unsigned)((h->n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, h->n,
                       h->d_row_ptr, h->d_w, h->d_U, h->d_K, work);
    GEMHIP_CHECK(hipGetLastError());
    int32_t *d_hubs = nullptr; int64_t *d_hoff = nullptr; unsigned char *d_scr = nullptr;
    if (!h->hub_rows.empty()) {                        // hub rows: a workgroup each (the one-lane kernel skipped them)
        const size_t nh = h->hub_rows.size();
        const size_t entries = (size_t)h->hub_off.back();
        const size_t scr_bytes = entries * 24 + (entries / ALIAS_CHUNK + nh + 1) * 24;
        GEMHIP_CHECK(hipMalloc((void **)&d_hubs, nh * sizeof(int32_t)));
        GEMHIP_CHECK(hipMalloc((void **)&d_hoff, (nh + 1) * sizeof(int64_t)));
        GEMHIP_CHECK(hipMalloc((void **)&d_scr, scr_bytes));
        GEMHIP_CHECK(hipMemcpyAsync(d_hubs, h->hub_rows.data(), nh * sizeof(int32_t), hipMemcpyHostToDevice, (hipStream_t)stream));
        GEMHIP_CHECK(hipMemcpyAsync(d_hoff, h->hub_off.data(), (nh + 1) * sizeof(int64_t), hipMemcpyHostToDevice, (hipStream_t)stream));
        hipLaunchKernelGGL(n2v_alias_hub_kernel, dim3((unsigned)nh), dim3(256), 0, (hipStream_t)stream, d_hubs, d_hoff, h->d_row_ptr, h->d_w, h->d_U, h->d_K, d_scr);
        GEMHIP_CHECK(hipGetLastError());
    }
    GEMHIP_CHECK(hipStreamSynchronize((hipStream_t)stream));
    GEMHIP_CHECK(hipFree(work));
    hipFree(d_hubs); hipFree(d_hoff); hipFree(d_scr);
    return GEMHIP_OK;
}

extern "C" int gemhip_n2v_get_alias(gemhip_n2v_t h, float *U_host, int32_t *K_host, int32_t *col_sorted_host)
{
    GEMHIP_REQUIRE(h, "n2v_get_alias: NULL handle");
    if (col_sorted_host && h->nnz) GEMHIP_CHECK(hipMemcpy(col_sorted_host, h->d_col, h->nnz * sizeof(int32_t), hipMemcpyDeviceToHost));
    if (h->uniform_rows) return 1;          // no tables: rows are uniform
    GEMHIP_REQUIRE(h->d_U, "n2v_get_alias: build_alias not called");
    if (U_host) GEMHIP_CHECK(hipMemcpy(U_host, h->d_U, h->nnz * sizeof(float), hipMemcpyDeviceToHost));
    if (K_host) GEMHIP_CHECK(hipMemcpy(K_host, h->d_K, h->nnz * sizeof(int32_t), hipMemcpyDeviceToHost));
    return GEMHIP_OK;
}

static int ensure_walk_buffer(gemhip_n2v_t h, int64_t nwalks, int32_t walk_len)
{
    const int64_t need = nwalks * walk_len;
    if (need > h->walks_cap) {
        hipFree(h->d_walks); h->d_walks = nullptr; h->walks_cap = 0;
        GEMHIP_CHECK(hipMalloc((void **)&h->d_walks, std::max<int64_t>(need, 4) * sizeof(int32_t)));
        h->walks_cap = need;
    }
    h->nwalks = nwalks; h->walk_len = walk_len;
    return GEMHIP_OK;
}

extern "C" int gemhip_n2v_walks(gemhip_n2v_t h, float p, float q, int32_t num_walks, int32_t walk_len, uint64_t seed,
                                int32_t flags, int64_t walk_begin, int64_t walk_end, void *stream)
{
    GEMHIP_REQUIRE(h, "n2v_walks: NULL handle");
    GEMHIP_REQUIRE(p > 0.f && q > 0.f, "n2v_walks: p=%g q=%g must be > 0", (double)p, (double)q);
    GEMHIP_REQUIRE(num_walks >= 1 && walk_len >= 1 && walk_len < 65536, "n2v_walks: num_walks=%d walk_len=%d", num_walks, walk_len);
    const int64_t total = h->m_start * (int64_t)num_walks;
    GEMHIP_REQUIRE(0 <= walk_begin && walk_begin <= walk_end && walk_end <= total, "n2v_walks: bad walk range [%lld,%lld) of %lld",
                   (long long)walk_begin, (long long)walk_end, (long long)total);
    if (!h->uniform_rows && !h->d_U) { if (int rc = gemhip_n2v_build_alias(h, stream)) return rc; }
    const int64_t count = walk_end - walk_begin;
    if (int rc = ensure_walk_buffer(h, count, walk_len)) return rc;
    h->walk_id_offset = walk_begin;
    h->unigram_ready = false; h->hotkey_state = 0;
    if (count == 0) return GEMHIP_OK;
    const bool second = !(p == 1.0f && q == 1.0f);
    const float ip = 1.0f / p, iq = 1.0f / q;
    const float amax = std::max(1.0f, std::max(ip, iq));
    const dim3 grid((unsigned)((count + 255) / 256)), block(256);
    hipStream_t s = (hipStream_t)stream;
    const uint32_t hb = half_bits((uint64_t)h->m_start);
#define N2V_WALK(S, W) hipLaunchKernelGGL((n2v_walk_kernel<S, W>), grid, block, 0, s, h->n, h->m_start, h->d_start, hb, h->d_row_ptr, h->d_col, h->d_U, h->d_K, ip, \
                                          iq, amax, walk_len, seed, flags, walk_begin, count, h->d_walks)
    if (second) { if (h->uniform_rows) N2V_WALK(true, false); else N2V_WALK(true, true); }
    else        { if (h->uniform_rows) N2V_WALK(false, false); else N2V_WALK(false, true); }
#undef N2V_WALK
    GEMHIP_CHECK(hipGetLastError());
    return GEMHIP_OK;
}

extern "C" int gemhip_n2v_start_nodes(gemhip_n2v_t h, int64_t *m)
{
    GEMHIP_REQUIRE(h && m, "n2v_start_nodes: NULL argument");
    *m = h->m_start;
    return GEMHIP_OK;
}

extern "C" int gemhip_n2v_set_walks(gemhip_n2v_t h, const int32_t *walks_host, int64_t nwalks, int32_t walk_len, int64_t walk_id_offset)
{
    GEMHIP_REQUIRE(h && walks_host && nwalks >= 0 && walk_len >= 1 && walk_len < 65536, "n2v_set_walks: bad arguments");
    if (int rc = ensure_walk_buffer(h, nwalks, walk_len)) return rc;
    h->walk_id_offset = walk_id_offset;
    h->unigram_ready = false; h->hotkey_state = 0;
    if (nwalks) GEMHIP_CHECK(hipMemcpy(h->d_walks, walks_host, nwalks * walk_len * sizeof(int32_t), hipMemcpyHostToDevice));
    return GEMHIP_OK;
}

extern "C" int gemhip_n2v_get_walks(gemhip_n2v_t h, int32_t *walks_host)
{
    GEMHIP_REQUIRE(h && walks_host, "n2v_get_walks: NULL argument");
    GEMHIP_CHECK(hipDeviceSynchronize());
    if (h->nwalks) GEMHIP_CHECK(hipMemcpy(walks_host, h->d_walks, h->nwalks * h->walk_len * sizeof(int32_t), hipMemcpyDeviceToHost));
    return GEMHIP_OK;
}

extern "C" int gemhip_n2v_walks_ptr(gemhip_n2v_t h, void **d_walks, int64_t *nwalks, int32_t *walk_len)
{
    GEMHIP_REQUIRE(h && d_walks, "n2v_walks_ptr: NULL argument");
    *d_walks = h->d_walks;
    if (nwalks) *nwalks = h->nwalks;
    if (walk_len) *walk_len = h->walk_len;
    return GEMHIP_OK;
}

extern "C" int gemhip_n2v_vocab(gemhip_n2v_t h, void *stream)
{
    GEMHIP_REQUIRE(h, "n2v_vocab: NULL handle");
    hipStream_t s = (hipStream_t)stream;
    GEMHIP_CHECK(hipMemsetAsync(h->d_counts, 0, h->n * sizeof(int32_t), s));
    const int64_t ntok = h->nwalks * h->walk_len;
    if (ntok) {
        const int64_t blocks = std::min<int64_t>((ntok + 255) / 256, 256 * 16);
        hipLaunchKernelGGL(n2v_vocab_kernel, dim3((unsigned)blocks), dim3(256), 0, s, h->d_walks, ntok, h->d_counts);
        GEMHIP_CHECK(hipGetLastError());
    }
    h->unigram_ready = false; h->hotkey_state = 0;
    return GEMHIP_OK;
}

extern "C" int gemhip_n2v_bind_counts(gemhip_n2v_t h, void *d_counts)
{
    GEMHIP_REQUIRE(h && d_counts, "n2v_bind_counts: NULL argument");
    if (h->own_counts) hipFree(h->d_counts);
    h->d_counts = (int32_t *)d_counts; h->own_counts = false; h->unigram_ready = false; h->hotkey_state = 0;
    return GEMHIP_OK;
}

extern "C" int gemhip_sgns_pairs(gemhip_n2v_t h, int64_t *pairs, int32_t reset)
{
    GEMHIP_REQUIRE(h && pairs, "sgns_pairs: NULL argument");
    GEMHIP_CHECK(hipDeviceSynchronize());
    unsigned long long v = 0;
    GEMHIP_CHECK(hipMemcpy(&v, h->d_pairs, sizeof v, hipMemcpyDeviceToHost));
    if (reset) GEMHIP_CHECK(hipMemset(h->d_pairs, 0, sizeof v));
    *pairs = (int64_t)v;
    return GEMHIP_OK;
}

extern "C" int gemhip_n2v_counts_ptr(gemhip_n2v_t h, void **d_counts)
{
    GEMHIP_REQUIRE(h && d_counts, "n2v_counts_ptr: NULL argument");
    *d_counts = h->d_counts;
    return GEMHIP_OK;
}


// InitUnigramTable (ELF @0x40e520) on a count vector: count^0.75, Vose in fp64, stacks popped from the back.
static bool vose_unigram(const int32_t *cnt, int64_t n, int64_t stride, std::vector<float> &Uf, std::vector<int32_t> &K)
{
    std::vector<double> U(n);
    std::vector<int32_t> small, large;
    K.assign(n, 0);
    double total = 0.0;
    for (int64_t i = 0; i < n; ++i) { U[i] = std::pow((double)cnt[i * stride], 0.75); total += U[i]; }
    if (!(total > 0.0)) return false;
    for (int64_t i = 0; i < n; ++i) U[i] /= total;
    small.reserve(n); large.reserve(n);
    for (int64_t i = 0; i < n; ++i) {
        U[i] = U[i] * (double)n;
        if (U[i] < 1.0) small.push_back((int32_t)i); else large.push_back((int32_t)i);
    }
    while (!small.empty() && !large.empty()) {
        const int32_t s = small.back(); small.pop_back();
        const int32_t l = large.back(); large.pop_back();
        K[s] = l;
        U[l] = U[l] + U[s] - 1.0;
        if (U[l] < 1.0) small.push_back(l); else large.push_back(l);
    }
    for (int32_t s : small) U[s] = 1.0;
    for (int32_t l : large) U[l] = 1.0;
    Uf.resize(n);
    for (int64_t i = 0; i < n; ++i) Uf[i] = (float)U[i];
    return true;
}

extern "C" int gemhip_n2v_build_unigram(gemhip_n2v_t h, int32_t *counts_out, float *UT_out, int32_t *KT_out)
{
    GEMHIP_REQUIRE(h, "n2v_build_unigram: NULL handle");
    GEMHIP_CHECK(hipDeviceSynchronize());
    const int64_t n = h->n;
    std::vector<int32_t> cnt(n), K;
    std::vector<float> Uf;
    { PhaseScope ph(PH_D2H); GEMHIP_CHECK(hipMemcpy(cnt.data(), h->d_counts, n * sizeof(int32_t), hipMemcpyDeviceToHost)); }
    {
        PhaseScope ph(PH_HOST);
        GEMHIP_REQUIRE(vose_unigram(cnt.data(), n, 1, Uf, K), "n2v_build_unigram: empty vocabulary (no walks?)");
        h->vs.build(cnt.data(), (int64_t)cnt.size());
    }
    PhaseScope ph_up(PH_H2D);
    if (!h->d_UT) GEMHIP_CHECK(hipMalloc((void **)&h->d_UT, n * sizeof(float)));
    if (!h->d_KT) GEMHIP_CHECK(hipMalloc((void **)&h->d_KT, n * sizeof(int32_t)));
    GEMHIP_CHECK(hipMemcpy(h->d_UT, Uf.data(), n * sizeof(float), hipMemcpyHostToDevice));
    GEMHIP_CHECK(hipMemcpy(h->d_KT, K.data(), n * sizeof(int32_t), hipMemcpyHostToDevice));
    {
        std::vector<uint2> UK((size_t)n);
        for (int64_t i = 0; i < n; ++i) { uint32_t ub; memcpy(&ub, &Uf[i], 4); UK[i] = make_uint2(ub, (uint32_t)K[i]); }
        if (!h->d_UK) GEMHIP_CHECK(hipMalloc((void **)&h->d_UK, n * sizeof(uint2)));
        GEMHIP_CHECK(hipMemcpy(h->d_UK, UK.data(), n * sizeof(uint2), hipMemcpyHostToDevice));
    }
    h->unigram_ready = true; h->vocab_order = false; h->sk_state = -1; h->hotkey_state = 0;
    if (counts_out) std::copy(cnt.begin(), cnt.end(), counts_out);
    if (UT_out) std::copy(Uf.begin(), Uf.end(), UT_out);
    if (KT_out) std::copy(K.begin(), K.end(), KT_out);
    return GEMHIP_OK;
}

// Nodes in order of first appearance in a token buffer on the device (-1 tokens are padding): back[r] = node with the r-th smallest first token index
// (nodes that never occur sort last), cnt = the handle's token counts.  LearnVocab's renaming (ELF @0x40d560) is r -> back[r] over the nodes that occur.
static int first_appearance_order(gemhip_n2v_t h, const int32_t *d_tokens, int64_t ntok, std::vector<int32_t> &back, std::vector<int32_t> &cnt)
{
    // like its sibling builders: the walks / counts may have been produced on a non-blocking stream of the caller (staged C-ABI use); the launches
    // below go to the null stream, which does not wait for such a stream
    GEMHIP_CHECK(hipDeviceSynchronize());
    const int64_t n = h->n;
    if (!h->d_first) GEMHIP_CHECK(hipMalloc((void **)&h->d_first, n * sizeof(unsigned long long)));
    GEMHIP_CHECK(hipMemset(h->d_first, 0xff, n * sizeof(unsigned long long)));
    hipLaunchKernelGGL(n2v_first_token_kernel, dim3((unsigned)std::min<int64_t>((ntok + 255) / 256, 256 * 16)), dim3(256), 0, 0, d_tokens, ntok, h->d_first);
    GEMHIP_CHECK(hipGetLastError());
    // nodes sorted by their first token index on the device (radix sort of (first, node) pairs: nodes that never occur sort last)
    back.assign(n, 0); cnt.assign(n, 0);
    unsigned long long *keys_out = nullptr; int32_t *ids = nullptr, *ids_out = nullptr; void *tmp = nullptr; size_t tmp_bytes = 0;
    hipError_t e = hipMalloc((void **)&keys_out, n * sizeof(unsigned long long));          // (every exit below frees all three temporaries)
    if (e == hipSuccess) e = hipMalloc((void **)&ids, 2 * n * sizeof(int32_t));
    ids_out = ids + n;
    if (e == hipSuccess) { hipLaunchKernelGGL(iota_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, ids, n); e = hipGetLastError(); }
    if (e == hipSuccess) e = hipcub::DeviceRadixSort::SortPairs(nullptr, tmp_bytes, h->d_first, keys_out, ids, ids_out, (int)n);
    if (e == hipSuccess) e = hipMalloc(&tmp, std::max<size_t>(tmp_bytes, 16));
    if (e == hipSuccess) e = hipcub::DeviceRadixSort::SortPairs(tmp, tmp_bytes, h->d_first, keys_out, ids, ids_out, (int)n);
    if (e == hipSuccess) e = hipDeviceSynchronize();
    if (e == hipSuccess) { PhaseScope ph(PH_D2H); e = hipMemcpy(back.data(), ids_out, n * sizeof(int32_t), hipMemcpyDeviceToHost);
                           if (e == hipSuccess) e = hipMemcpy(cnt.data(), h->d_counts, n * sizeof(int32_t), hipMemcpyDeviceToHost); }
    hipFree(keys_out); hipFree(ids); hipFree(tmp);
    if (e != hipSuccess) return fail(GEMHIP_E_HIP, "first_appearance_order: device sort failed: %s", hipGetErrorString(e));
    return GEMHIP_OK;
}

// InitUnigramTable in the BINARY's layout.  LearnEmbeddings (ELF @0x40ea30) renames the tokens 0..N-1 in order of first appearance in the walk matrix
// (LearnVocab @0x40d560) and builds the alias table over those N entries in that order; RndUnigramInt (@0x40d5f0) then draws a SLOT floor(u N) of that
// table.  Vose's stack discipline makes the table depend on the order of its entries, so the node-id layout of gemhip_n2v_build_unigram is the same
// distribution but not the same table (measured on SBM-1024 with the binary-pinned restatement: -0.33 +- 0.24 % of MAP, profiles/r03_unigram_layout_effect.json).
// Here: first token index per node on the device (atomicMin over the walks), nodes that occur sorted by it (N = their number), Vose over their counts in
// that order, and the result stored in NODE space so that the kernels need no renaming of tokens or tables:
//   KTslot[slot] = node of KTable'[slot]   (flags & 2, the RndUnigramInt quirk; else node of slot)        -- what A.KT points at
//   UK[v]        = {UTable'[r(v)], node of KTable'[r(v)]}   with r(v) the renamed id of node v            -- what A.UK points at
// order_out[N]: node ids in first-appearance order; UT_out / KT_out [N]: the table in RENAMED indices, as the binary holds it (tests compare them
// with oracle/snap_stream.py's).  The walks must be the whole corpus of the handle (single-GPU path).
extern "C" int gemhip_n2v_build_unigram_vocab_order(gemhip_n2v_t h, int32_t flags, int64_t *n_vocab_out, int32_t *order_out, float *UT_out, int32_t *KT_out)
{
    GEMHIP_REQUIRE(h && h->d_walks && h->nwalks > 0, "n2v_build_unigram_vocab_order: no walks");
    const int64_t n = h->n;
    std::vector<int32_t> back, cnt;              // back: renamed id -> node
    if (int rc = first_appearance_order(h, h->d_walks, h->nwalks * h->walk_len, back, cnt)) return rc;
    PhaseScope ph_host(PH_HOST);
    int64_t N = 0;
    while (N < n && cnt[back[N]] > 0) ++N;             // the nodes that occur come first (a node occurs iff it has a first token iff its count > 0)
    back.resize(N);
    GEMHIP_REQUIRE(N > 0, "n2v_build_unigram_vocab_order: empty vocabulary");
    std::vector<int32_t> cr(N), K;
    std::vector<float> Uf;
    for (int64_t r = 0; r < N; ++r) cr[r] = cnt[back[r]];
    GEMHIP_REQUIRE(vose_unigram(cr.data(), N, 1, Uf, K), "n2v_build_unigram_vocab_order: empty vocabulary");
    h->vs.build(cnt.data(), n);
    std::vector<int32_t> slot(N);
    std::vector<uint2> UK((size_t)n, make_uint2(0u, 0u));          // nodes that never occur are never drawn
    for (int64_t r = 0; r < N; ++r) {
        slot[r] = (flags & 2) ? back[K[r]] : back[r];
        uint32_t ub; memcpy(&ub, &Uf[r], 4);
        UK[back[r]] = make_uint2(ub, (uint32_t)back[K[r]]);
    }
    if (!h->d_KTslot) GEMHIP_CHECK(hipMalloc((void **)&h->d_KTslot, n * sizeof(int32_t)));
    if (!h->d_UK) GEMHIP_CHECK(hipMalloc((void **)&h->d_UK, n * sizeof(uint2)));
    { PhaseScope ph(PH_H2D);
      GEMHIP_CHECK(hipMemcpy(h->d_KTslot, slot.data(), N * sizeof(int32_t), hipMemcpyHostToDevice));
      GEMHIP_CHECK(hipMemcpy(h->d_UK, UK.data(), n * sizeof(uint2), hipMemcpyHostToDevice)); }
    h->n_vocab = N; h->vocab_order = true; h->unigram_ready = true; h->sk_state = -1; h->hotkey_state = 0;
    if (n_vocab_out) *n_vocab_out = N;
    if (order_out) std::copy(back.begin(), back.end(), order_out);
    if (UT_out) std::copy(Uf.begin(), Uf.end(), UT_out);
    if (KT_out) std::copy(K.begin(), K.end(), KT_out);
    return GEMHIP_OK;
}

// One alias table per partition p = {v : v % parts == p} over local indices v / parts (restricted unigram).
extern "C" int gemhip_n2v_build_unigram_parts(gemhip_n2v_t h, int32_t parts, float *UT_out, int32_t *KT_out)
{
    GEMHIP_REQUIRE(h && parts >= 1 && parts <= h->n, "n2v_build_unigram_parts: bad arguments");
    GEMHIP_CHECK(hipDeviceSynchronize());
    const int64_t n = h->n;
    std::vector<int32_t> cnt(n);
    GEMHIP_CHECK(hipMemcpy(cnt.data(), h->d_counts, n * sizeof(int32_t), hipMemcpyDeviceToHost));
    h->vs.build(cnt.data(), (int64_t)cnt.size());
    std::vector<float> Uall(n);
    std::vector<int32_t> Kall(n);
    h->part_off.assign(parts + 1, 0);
    h->vs_part.assign(parts, VocabStats());
    std::vector<int32_t> cp;
    for (int32_t p = 0; p < parts; ++p) {
        const int64_t np = (n - p + parts - 1) / parts;
        h->part_off[p + 1] = h->part_off[p] + np;
        std::vector<float> Uf; std::vector<int32_t> K;
        GEMHIP_REQUIRE(vose_unigram(cnt.data() + p, np, parts, Uf, K), "n2v_build_unigram_parts: partition %d has an empty vocabulary", p);
        std::copy(Uf.begin(), Uf.end(), Uall.begin() + h->part_off[p]);
        std::copy(K.begin(), K.end(), Kall.begin() + h->part_off[p]);
        cp.resize(np);
        for (int64_t i = 0; i < np; ++i) cp[i] = cnt[p + i * parts];
        h->vs_part[p].build(cp.data(), np);
    }
    hipFree(h->d_UTp); hipFree(h->d_KTp); hipFree(h->d_UKp); h->d_UTp = nullptr; h->d_KTp = nullptr; h->d_UKp = nullptr;
    GEMHIP_CHECK(hipMalloc((void **)&h->d_UTp, n * sizeof(float)));
    GEMHIP_CHECK(hipMalloc((void **)&h->d_KTp, n * sizeof(int32_t)));
    GEMHIP_CHECK(hipMalloc((void **)&h->d_UKp, n * sizeof(uint2)));
    GEMHIP_CHECK(hipMemcpy(h->d_UTp, Uall.data(), n * sizeof(float), hipMemcpyHostToDevice));
    GEMHIP_CHECK(hipMemcpy(h->d_KTp, Kall.data(), n * sizeof(int32_t), hipMemcpyHostToDevice));
    {
        std::vector<uint2> UK((size_t)n);
        for (int64_t i = 0; i < n; ++i) { uint32_t ub; memcpy(&ub, &Uall[i], 4); UK[i] = make_uint2(ub, (uint32_t)Kall[i]); }
        GEMHIP_CHECK(hipMemcpy(h->d_UKp, UK.data(), n * sizeof(uint2), hipMemcpyHostToDevice));
    }
    h->parts = parts; h->skp_state = -1; h->parts_vocab_order = false;
    hipFree(h->d_SKp); h->d_SKp = nullptr;
    if (UT_out) std::copy(Uall.begin(), Uall.end(), UT_out);
    if (KT_out) std::copy(Kall.begin(), Kall.end(), KT_out);
    return GEMHIP_OK;
}

// The per-partition tables in the BINARY's layout (flags bit GEMHIP_N2V_VOCAB_ORDER on the N-GPU schedule): partition p = the nodes v % parts == p that
// occur, in order of first appearance in the whole corpus -- d_corpus: the walks of ALL ranks in walk-id order (rank-major shards, -1 tokens = padding;
// NULL = this handle's own walks: one rank) -- Vose over their counts (the handle's counts: summed over the ranks) in that order; the slot table and the
// alias arrays are stored by LOCAL row v / parts like the node-id layout's.  With one partition this IS gemhip_n2v_build_unigram_vocab_order's table.
// Optional outputs: UT_out / KT_out [n] by local row (concatenated partitions, as d_UTp / d_KTp), slot_out [n] (partition p's slots at part_off[p],
// -1 beyond its slot count), nslots_out [parts].
extern "C" int gemhip_n2v_build_unigram_parts_vocab_order(gemhip_n2v_t h, int32_t parts, int32_t flags, const void *d_corpus, int64_t corpus_tokens,
                                                          float *UT_out, int32_t *KT_out, int32_t *slot_out, int64_t *nslots_out)
{
    GEMHIP_REQUIRE(h && parts >= 1 && parts <= h->n, "n2v_build_unigram_parts_vocab_order: bad arguments");
    GEMHIP_REQUIRE(d_corpus ? corpus_tokens > 0 : (h->d_walks && h->nwalks > 0), "n2v_build_unigram_parts_vocab_order: no walks");
    const int64_t n = h->n;
    std::vector<int32_t> back, cnt;
    if (int rc = first_appearance_order(h, d_corpus ? (const int32_t *)d_corpus : h->d_walks, d_corpus ? corpus_tokens : h->nwalks * h->walk_len, back, cnt)) return rc;
    int64_t N = 0;
    while (N < n && cnt[back[N]] > 0) ++N;
    GEMHIP_REQUIRE(N > 0, "n2v_build_unigram_parts_vocab_order: empty vocabulary");
    h->vs.build(cnt.data(), n);
    h->part_off.assign(parts + 1, 0);
    h->vs_part.assign(parts, VocabStats());
    h->part_slots.assign(parts, 0);
    for (int32_t p = 0; p < parts; ++p) h->part_off[p + 1] = h->part_off[p] + (n - p + parts - 1) / parts;
    std::vector<float> Uall(n, 0.f);
    std::vector<int32_t> Kall(n, 0), Sall(n, -1), cp;
    std::vector<std::vector<int32_t>> L(parts);                // partition p's occurring nodes in first-appearance order
    for (int64_t r = 0; r < N; ++r) L[back[r] % parts].push_back(back[r]);
    for (int32_t p = 0; p < parts; ++p) {
        const int64_t off = h->part_off[p], np = h->part_off[p + 1] - off, Np = (int64_t)L[p].size();
        GEMHIP_REQUIRE(Np > 0, "n2v_build_unigram_parts_vocab_order: partition %d has an empty vocabulary", p);
        std::vector<int32_t> cr(Np), K; std::vector<float> Uf;
        for (int64_t r = 0; r < Np; ++r) cr[r] = cnt[L[p][r]];
        GEMHIP_REQUIRE(vose_unigram(cr.data(), Np, 1, Uf, K), "n2v_build_unigram_parts_vocab_order: partition %d has an empty vocabulary", p);
        for (int64_t r = 0; r < Np; ++r) {
            const int32_t loc = L[p][r] / parts, ali = L[p][K[r]] / parts;
            Uall[off + loc] = Uf[r]; Kall[off + loc] = ali;
            Sall[off + r] = (flags & 2) ? ali : loc;
        }
        h->part_slots[p] = Np;
        cp.resize(np);
        for (int64_t i = 0; i < np; ++i) cp[i] = cnt[p + i * parts];
        h->vs_part[p].build(cp.data(), np);
    }
    hipFree(h->d_UTp); hipFree(h->d_KTp); hipFree(h->d_UKp); hipFree(h->d_KTslotp); h->d_UTp = nullptr; h->d_KTp = nullptr; h->d_UKp = nullptr; h->d_KTslotp = nullptr;
    GEMHIP_CHECK(hipMalloc((void **)&h->d_UTp, n * sizeof(float)));
    GEMHIP_CHECK(hipMalloc((void **)&h->d_KTp, n * sizeof(int32_t)));
    GEMHIP_CHECK(hipMalloc((void **)&h->d_UKp, n * sizeof(uint2)));
    GEMHIP_CHECK(hipMalloc((void **)&h->d_KTslotp, n * sizeof(int32_t)));
    {
        PhaseScope ph(PH_H2D);
        std::vector<uint2> UK((size_t)n);
        for (int64_t i = 0; i < n; ++i) { uint32_t ub; memcpy(&ub, &Uall[i], 4); UK[i] = make_uint2(ub, (uint32_t)Kall[i]); }
        std::vector<int32_t> Sdev(Sall);
        for (auto &v : Sdev) if (v < 0) v = 0;                 // (never read: a partition's launches draw slots below its slot count)
        GEMHIP_CHECK(hipMemcpy(h->d_UTp, Uall.data(), n * sizeof(float), hipMemcpyHostToDevice));
        GEMHIP_CHECK(hipMemcpy(h->d_KTp, Kall.data(), n * sizeof(int32_t), hipMemcpyHostToDevice));
        GEMHIP_CHECK(hipMemcpy(h->d_UKp, UK.data(), n * sizeof(uint2), hipMemcpyHostToDevice));
        GEMHIP_CHECK(hipMemcpy(h->d_KTslotp, Sdev.data(), n * sizeof(int32_t), hipMemcpyHostToDevice));
    }
    h->parts = parts; h->skp_state = -1; h->parts_vocab_order = true;
    hipFree(h->d_SKp); h->d_SKp = nullptr;
    if (UT_out) std::copy(Uall.begin(), Uall.end(), UT_out);
    if (KT_out) std::copy(Kall.begin(), Kall.end(), KT_out);
    if (slot_out) std::copy(Sall.begin(), Sall.end(), slot_out);
    if (nslots_out) std::copy(h->part_slots.begin(), h->part_slots.end(), nslots_out);
    return GEMHIP_OK;
}

extern "C" int gemhip_sgns_init(gemhip_n2v_t h, int32_t d, uint64_t seed, void *dSynPos, void *dSynNeg)
{
    GEMHIP_REQUIRE(h && d >= 1, "sgns_init: bad arguments");
    GEMHIP_REQUIRE(pick_sgns(d) != nullptr, "sgns_init: d=%d unsupported (even d <= 512, odd d <= 256)", d);
    GEMHIP_REQUIRE((dSynPos == nullptr) == (dSynNeg == nullptr), "sgns_init: pass both or neither external table");
    if (h->own_syn) { hipFree(h->SynPos); hipFree(h->SynNeg); h->own_syn = false; }
    const size_t bytes = (size_t)h->n * d * sizeof(float);
    if (dSynPos) { h->SynPos = (float *)dSynPos; h->SynNeg = (float *)dSynNeg; }
    else {
        GEMHIP_CHECK(hipMalloc((void **)&h->SynPos, bytes));
        GEMHIP_CHECK(hipMalloc((void **)&h->SynNeg, bytes));
        h->own_syn = true;
    }
    h->d = d;
    const int64_t total = h->n * (int64_t)d;
    const int64_t threads = (total + 3) / 4;
    hipLaunchKernelGGL(sgns_init_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, 0, h->SynPos, h->SynNeg, total, d, seed);
    GEMHIP_CHECK(hipGetLastError());
    return GEMHIP_OK;
}

extern "C" int gemhip_n2v_set_max_waves(gemhip_n2v_t h, int32_t max_waves)
{
    GEMHIP_REQUIRE(h && max_waves >= 0, "n2v_set_max_waves: bad arguments");
    h->kn.max_waves = max_waves;
    return GEMHIP_OK;
}

extern "C" int gemhip_sgns_set_tables(gemhip_n2v_t h, const float *SynPos_host, const float *SynNeg_host)
{
    GEMHIP_REQUIRE(h && h->SynPos && SynPos_host && SynNeg_host, "sgns_set_tables: bad arguments (call sgns_init first)");
    const size_t bytes = (size_t)h->n * h->d * sizeof(float);
    GEMHIP_CHECK(hipMemcpy(h->SynPos, SynPos_host, bytes, hipMemcpyHostToDevice));
    GEMHIP_CHECK(hipMemcpy(h->SynNeg, SynNeg_host, bytes, hipMemcpyHostToDevice));
    return GEMHIP_OK;
}

extern "C" int gemhip_sgns_get_tables(gemhip_n2v_t h, float *SynPos_host, float *SynNeg_host)
{
    GEMHIP_REQUIRE(h && h->SynPos, "sgns_get_tables: no tables");
    GEMHIP_CHECK(hipDeviceSynchronize());
    const size_t bytes = (size_t)h->n * h->d * sizeof(float);
    PhaseScope ph(PH_D2H);
    if (SynPos_host) GEMHIP_CHECK(hipMemcpy(SynPos_host, h->SynPos, bytes, hipMemcpyDeviceToHost));
    if (SynNeg_host) GEMHIP_CHECK(hipMemcpy(SynNeg_host, h->SynNeg, bytes, hipMemcpyDeviceToHost));
    return GEMHIP_OK;
}

// Which kernel, how many concurrent wavefronts and which rows are "hot" for one pass of TrainModel over `nwalks` walks.  Pure host arithmetic.
//
// Concurrency against quality -- the rule and where it comes from.  What Hogwild costs here is LOST UPDATES: every wavefront
// keeps its 5 negative rows per pair open from their load to their store; a store another wavefront makes to such a row in between is
// overwritten.  The CPU replay of this kernel's concurrency (scripts/hogwild_emul: W virtual wavefronts, the kernel's private copies) puts
// ~90 % of the MAP loss on those overwritten negative-row updates, ~10 % on the centre row's, none on stale gradients; the expected fraction
// of overwriting stores is  rho = W x 5 x w / n_eff  with w = the window in pair steps (prefetch + 1 without RELOAD; ~0.4 with it: one reload
// round trip, ~0.5 us against a 1.35 us step) and n_eff = 1 / sum_v q_v^2 the EFFECTIVE table size of the negative-sampling distribution
// (unigram^0.75): n on a graph whose nodes are equally frequent (SBM: n / 1.06), far smaller on a power-law graph (R-MAT scale 17: 11 316 of
// 131 072 nodes).  Measured at SBM 1M/10M against the sequential oracle (same seed, paired per-node AP): without RELOAD rho = 1.1 % / 1.5 % /
// 2.3 % (768 / 1024 / 1536 wavefronts) cost -0.1 % / -0.4..-0.9 % / -0.8..-1.3 % of MAP, and 1536 wavefronts with prefetch 1 (rho 1.5 %)
// -0.5 %: the loss follows rho, not the wavefront count; with RELOAD 1536 wavefronts (rho 0.3 %) measure +0.15 +- 0.25 %.  Default: rho <= 1.5 %.
static SgnsLaunchPlan plan_sgns_launch(const VocabStats &vs, const SgnsKnobs &kn, int64_t n, int32_t d, int32_t window, int32_t walk_len,
                                       int64_t nwalks, int32_t flags)
{
    SgnsLaunchPlan P;
    const bool deterministic = (flags & 4) != 0;
    // window-cached kernel (default): radius R = tokens either side of the centre whose SynPos row stays in LDS
    int R = kn.cache_radius < 0 ? 10 : kn.cache_radius;
    R = std::min(R, std::min(window, 31));
    const int rw = sgns_win_row_floats(d);
    const size_t ints = (size_t)((walk_len + 4 * window * SGNS_NEG + 3) & ~3);
    // tokens + negative targets of two centres, the window rows (twice with the delta write-back: as trained / as loaded) and -- unless every context
    // row is cached (`allc`: R >= window and no hot rows, the launcher's ALLC instantiations) -- one staging row for an uncached context
    // (kn.part: the bucket kernels keep the as-loaded copies in a global scratch instead of LDS)
    auto lds_bytes = [&](bool delta, bool allc) { return ints * sizeof(int32_t) + (size_t)((2 * R + 1) * ((delta && !kn.part) ? 2 : 1) + (allc ? 0 : 1)) * rw * sizeof(float); };
    // the window has to fit a block's LDS: wide rows / long walks fall back to sgns_kernel
    P.window = R > 0 && !(flags & GEMHIP_N2V_NO_WINDOW_CACHE) && 2 * window * SGNS_NEG <= 2 * WAVE && walk_len >= 2 && lds_bytes(true, false) <= 64 * 1024 &&
               (d % 2 == 0 ? d <= 512 : d <= 256);
    if (!P.window) {
        // sgns_kernel: four wavefronts per workgroup, no private copies beyond the pair in flight.  Small graphs: when the open rows approach n,
        // concurrent writers overwrite each other's updates and the embedding degrades (tests/test_n2v_gpu.py): n / 128 wavefronts
        const size_t per_wave = (size_t)(walk_len + 2 * window * SGNS_NEG) * sizeof(int32_t);
        if (!deterministic) {
            const int64_t hog_cap = kn.max_waves > 0 ? kn.max_waves : std::max<int64_t>(1, n / HOGWILD_ROWS_PER_WAVE);
            P.waves = std::min<int64_t>(std::min<int64_t>(hog_cap, 256 * 16), nwalks);          // 16 waves/CU already saturate the fabric (scripts/ab_sgns_waves.py)
            P.threads = 256; P.blocks = (int)((P.waves + 3) / 4);
        }
        P.lds = per_wave * (P.threads / 64);
        return P;
    }
    P.R = R;
    const int mode = kn.cache_delta;                 // -1 auto: delta write-back whenever other wavefronts train concurrently
    P.delta = deterministic && mode == 1;            // (cache_delta 1 on a deterministic launch: the Hogwild code path on ONE wavefront, for the parity tests)
    // a node expected to sit in another wavefront's window at any time -- (W - 1) x (2R + 1) x count / tokens >= 1 -- is hot
    const double span = kn.window_span > 0 ? (double)kn.window_span : (double)(2 * R + 1);
    auto hot_threshold = [&](int64_t waves) -> int32_t {
        if (kn.hot_count > 0) return kn.hot_count;
        if (kn.hot_count < 0 && waves > 1 && vs.total > 0.0) {
            const double thr = vs.total / ((double)(waves - 1) * span);
            if (vs.max >= thr) return (int32_t)std::max(2.0, std::ceil(thr));
        }
        return 0;
    };
    bool allc = R >= window && !kn.has_local_hot;
    if (!deterministic) {
        P.delta = mode != 0;
        // the (reload, prefetch) pair the launcher will actually run: launches WITHOUT reload-on-update and with prefetch distance 1 exist for the
        // benchmark shape only (d == 128, whole window cached, no hot rows -- launch_sgns_win's AB instantiations); every other Hogwild launch is
        // reload-on-update with prefetch distance 2 whatever the knobs say, and the width rule has to be computed for THAT kernel
        auto w_steps_of = [&](bool all_cached) -> double {
            const bool ab_shape = d == 128 && all_cached;
            const bool reload_eff = P.delta && (kn.reload || !ab_shape);
            return reload_eff ? 0.4 : (double)((ab_shape ? kn.prefetch : 2) + 1);
        };
        const double n_eff = vs.n_eff > 0.0 ? vs.n_eff : (double)n;
        // graphs below 8192 nodes: the window rows themselves (2R+1 per wavefront) are a sizeable part of the table -- SBM-1024 (d=16) loses
        // 3.5 % of MAP at 8 wavefronts and nothing at 2: bound the open fraction of the table at 1/16
        const int64_t hog_tiny = std::max<int64_t>(1, n / (16 * (8 + 2 * R + 1)));
        // ... but never more wavefronts than 2 % of the rows that occur at all: that is as far as the measurements behind this rule reach (stale
        // gradients cost nothing up to there -- CPU replay at 0.4 %, SBM 100k at 0.8 %, R-MAT scale 17 at 2.0 % of the active rows; R-MAT scale 13 with
        // 26 % of its 5 936 active rows open ended 21 % ABOVE the sequential algorithm's MAP: its hubs under-trained)
        const int64_t w_act = std::max<int64_t>(1, (int64_t)((vs.active > 0.0 ? vs.active : (double)n) / 50.0));
        // ... and, on graphs with hubs, never more than the CONCURRENT-TOUCH bound of round 5: (W - 1) x touch2_hub <= 0.165, touch2_hub = the hubs' part of
        // sum_v (p_v + 5 q_v)^2 (p = token share: the context; q = unigram^0.75 share: the five negatives) -- minus the 40 / active a table of equally frequent
        // rows has, so that SBM graphs stay on the rho rule they were validated on.  What round 6 found out about it (profiles/r06_*.jsonl):
        //  * the bound was fitted through launches whose gaps were dominated by a HEAVY TAIL that has nothing to do with hubs: an isolated edge trained by two
        //    wavefronts at once (LOCALLY HOT ROWS, ensure_hotkey below -- now handled; launches at one width agree to 0.2 % where they scattered over 6 %);
        //  * the staleness of the hubs' gradients is not the cause either: instrumented (GEMHIP_SGNS_STALENESS), the top hub's row takes 22 foreign updates
        //    between a wavefront's read and its add at 768 wavefronts; fresh reads / returning atomics cut that to 6 and the gap does not move; making every
        //    Hogwild write a lossless add makes it slightly WORSE; lowering the hot-row threshold 2-16x makes it worse and slower;
        //  * what is left is smooth in the width and larger on the larger graph -- R-MAT scale 17: -0.45 / -1.6 / -1.8 % at 256 / 768 / 1 536 wavefronts, scale 20:
        //    -1.3 / -3.0 / -6.0 % (paired with the sequential oracle, s.e. 0.3-0.5 %, 2-4 launches each) -- carried by query nodes of 1 000..30 000 tokens, and
        //    slower launches of the same configuration are the worse ones (queueing of the hot rows' atomic adds at the memory side is the suspect).
        // So the bound stays as the conservative extrapolation it is (548 wavefronts on scale 22 -- both oracle runs exist since round 6, their scoring did not finish: no pin yet), with a
        // FLOOR of 256 wavefronts: at 256 both measured graphs are at or inside -1.3 % (round 5's 50 wavefronts on scale 17 bought nothing for 3.5x the time:
        // ADVICE r5), and ONE bound for both unigram-table layouts -- round 5 halved it for the node-id layout on the strength of two launches (-6.3 % at 207
        // wavefronts) that the heavy tail explains; after the fix the layouts measure alike (scale 20, 768 wavefronts: -2.7 / -3.3 % and -2.5 / -4.3 %).
        const double touch_bound = 0.165;
        const double touch2_hub = std::max(0.0, vs.touch2 - 40.0 / std::max(1.0, vs.active > 0.0 ? vs.active : (double)n));
        const int64_t w_touch = touch2_hub > 0.0 ? std::max<int64_t>(256, 1 + (int64_t)std::min(1e15, touch_bound / (touch2_hub * kn.touch_scale))) : INT64_MAX;
        auto width = [&](bool all_cached) -> int64_t {
            // registers: the single-GPU kernels allocate 176-184 VGPRs (2 wavefronts per SIMD = 8 per CU), the bucket kernels 136-145 (3 per SIMD = 12 per CU)
            const int64_t per_cu = std::max<int64_t>(1, std::min<int64_t>(kn.part ? 12 : 8, (int64_t)(160 * 1024) / (int64_t)(lds_bytes(P.delta, all_cached) + 512)));
            const int64_t w_dev = std::min<int64_t>(256 * per_cu, nwalks);
            const double w_steps = w_steps_of(all_cached);
            const bool reload_eff = w_steps < 1.0;
            int64_t hog_rho = std::max<int64_t>(1, (int64_t)(0.015 * n_eff / (5.0 * w_steps * kn.duty)));
            // hot rows: with W wavefronts the nodes with count >= tokens / ((W-1)(2R+1)) stay out of the LDS windows and take their negative updates
            // by atomic add (sgns_win_kernel), so the rule only has to hold over the remaining (cold) rows: the largest W that satisfies it
            if (reload_eff && kn.hot_count < 0 && n >= 8192 && hog_rho < std::min(std::min(w_dev, w_act), w_touch) && vs.total > 0.0) {
                for (int64_t wtry = std::min(std::min(w_dev, w_act), w_touch); wtry > hog_rho; wtry = wtry * 7 / 8) {
                    const double thr = std::max(2.0, std::ceil(vs.total / ((double)(wtry - 1) * span)));
                    if (0.015 * vs.n_eff_cold(thr) / (5.0 * w_steps * kn.duty) >= (double)wtry) { hog_rho = wtry; break; }
                }
            }
            const int64_t hog_win = kn.max_waves > 0 ? kn.max_waves : n >= 8192 ? std::min(hog_rho, w_touch) : std::min(hog_rho, hog_tiny);
            int64_t w = std::min<int64_t>(hog_win, w_dev);
            // Speed only (round 5): hot rows are updated by atomic adds at the memory side, and those saturate long before the device's wavefront slots do:
            // past ~three wavefronts per CU more wavefronts only queue up behind the same rows and touch them more often at once.  Measured per SGNS launch,
            // same box each (profiles/r05_rmat22_width_sweep.jsonl, r05_rmat20_width_sweep.jsonl): R-MAT scale 22 41.3 / 33.0 / 34.3 / 36.7 s and scale 20
            // 12.5 / 10.3 / 10.7 / 15.0 s at 512 / 768 / 1024 / 1536 wavefronts.  (A rate model -- min(W / wave step, atomic row operations per second / hot
            // operations per pair) -- fitted scale 22 and 17 and then put scale 20 at 496 wavefronts, 25 % slower than 768: the atomic unit's rate is not a
            // constant of the device, it grows with the number of distinct hot rows.  Dropped for the plain cap.)  Widths only ever shrink here.
            if (kn.max_waves == 0 && !kn.part && reload_eff && kn.hot_count < 0 && n >= 8192 && w > 768 && vs.total > 0.0 &&
                vs.max >= std::max(2.0, std::ceil(vs.total / ((double)(w - 1) * span))))
                w = 768;
            return w;
        };
        // the launch without the staging row holds one more wavefront per CU at d = 128 -- but only exists when no row is hot AT THAT WIDTH
        P.waves = width(allc);
        if (allc && hot_threshold(P.waves) != 0) { allc = false; P.waves = width(false); }
        if (P.waves == 1 && mode < 0) P.delta = false;
    }
    P.hot_thr = hot_threshold(P.waves);
    if (P.hot_thr == 0 && kn.has_local_hot && P.delta && P.waves > 1) P.hot_thr = INT32_MAX;       // only the locally hot nodes (hotkey INT32_MAX) qualify
    // (the launcher takes an ALLC instantiation iff R >= window && hot_thr == 0: `allc` false with hot_thr 0 only gives that kernel a row it does not use)
    P.lds = lds_bytes(P.delta, allc && P.hot_thr == 0);
    P.blocks = (int)P.waves; P.threads = 64;
    return P;
}

// LOCALLY HOT ROWS: the array the Hogwild kernels compare with hot_thr -- the token count, or INT32_MAX for a node whose tokens are packed into few walks
// (kernels above).  Built on `stream` from the walks this handle holds and its (possibly externally reduced) counts; rebuilt after walks / vocabulary
// change.  n_local_hot is read back (one 4-byte copy): a launch without any such node and without count-hot rows keeps the all-cached instantiation.
static int ensure_hotkey(gemhip_n2v_t h, hipStream_t stream)
{
    if (h->hotkey_state == 1) return GEMHIP_OK;
    if (!h->d_hotkey) {
        GEMHIP_CHECK(hipMalloc((void **)&h->d_hotkey, (size_t)h->n * sizeof(int32_t)));
        GEMHIP_CHECK(hipMalloc((void **)&h->d_wcount, (size_t)h->n * sizeof(int32_t)));
        GEMHIP_CHECK(hipMalloc((void **)&h->d_nlocal, sizeof(unsigned int)));
    }
    GEMHIP_CHECK(hipMemsetAsync(h->d_wcount, 0, (size_t)h->n * sizeof(int32_t), stream));
    GEMHIP_CHECK(hipMemsetAsync(h->d_nlocal, 0, sizeof(unsigned int), stream));
    if (h->nwalks > 0) {
        const int64_t blocks = std::min<int64_t>(h->nwalks, 256 * 32);
        hipLaunchKernelGGL(n2v_walk_presence_kernel, dim3((unsigned)blocks), dim3(64), (size_t)h->walk_len * sizeof(int32_t), stream, h->d_walks, h->nwalks, h->walk_len, h->d_wcount);
    }
    hipLaunchKernelGGL(n2v_hotkey_kernel, dim3((unsigned)((h->n + 255) / 256)), dim3(256), 0, stream, h->n, h->d_counts, h->d_wcount, h->kn.local_hot, h->d_hotkey, h->d_nlocal);
    GEMHIP_CHECK(hipGetLastError());
    unsigned int nl = 0;
    GEMHIP_CHECK(hipMemcpyAsync(&nl, h->d_nlocal, sizeof nl, hipMemcpyDeviceToHost, stream));
    GEMHIP_CHECK(hipStreamSynchronize(stream));
    h->n_local_hot = nl;
    h->hotkey_state = 1;
    return GEMHIP_OK;
}

// The slot table the window kernels draw negatives from (SgnsArgs::SK), for the table currently held and the given quirk bit.  Built on `stream`,
// in front of the launch that needs it; rebuilt only after a table build or when the quirk bit changes.
static int ensure_slot_table(gemhip_n2v_t h, const int32_t *KT, int64_t nslots, int quirk, hipStream_t stream)
{
    if (h->sk_state == quirk && h->d_SK) return GEMHIP_OK;
    if (nslots > h->sk_cap) {
        if (h->d_SK) { GEMHIP_CHECK(hipDeviceSynchronize()); hipFree(h->d_SK); h->d_SK = nullptr; h->sk_cap = 0; }
        GEMHIP_CHECK(hipMalloc((void **)&h->d_SK, (size_t)nslots * sizeof(uint4)));
        h->sk_cap = nslots;
    }
    hipLaunchKernelGGL(n2v_slot_table_kernel, dim3((unsigned)((nslots + 255) / 256)), dim3(256), 0, stream, nslots, KT, h->d_UK, quirk, h->d_SK);
    GEMHIP_CHECK(hipGetLastError());
    h->sk_state = quirk;
    return GEMHIP_OK;
}
// ... and of the per-partition tables: partition p's slots occupy [part_off[p], part_off[p+1]) of d_SKp, entries in LOCAL indices like d_KTp / d_UKp
static int ensure_slot_table_parts(gemhip_n2v_t h, int quirk, hipStream_t stream)
{
    if (h->skp_state == quirk && h->d_SKp) return GEMHIP_OK;
    if (!h->d_SKp) GEMHIP_CHECK(hipMalloc((void **)&h->d_SKp, (size_t)h->n * sizeof(uint4)));
    for (int32_t p = 0; p < h->parts; ++p) {
        const int64_t off = h->part_off[p], np = h->parts_vocab_order ? h->part_slots[p] : h->part_off[p + 1] - off;
        if (np <= 0) continue;
        // (the binary's layout: the slot table names the entry -- quirk already folded in --, the alias arrays are indexed by local row)
        hipLaunchKernelGGL(n2v_slot_table_kernel, dim3((unsigned)((np + 255) / 256)), dim3(256), 0, stream, np, (h->parts_vocab_order ? h->d_KTslotp : h->d_KTp) + off,
                           h->d_UKp + off, h->parts_vocab_order ? 1 : quirk, h->d_SKp + off);
    }
    GEMHIP_CHECK(hipGetLastError());
    h->skp_state = quirk;
    return GEMHIP_OK;
}

extern "C" int gemhip_sgns_train(gemhip_n2v_t h, int32_t window, int32_t neg, float alpha0, int32_t epochs, int32_t epoch,
                                 int64_t walk_lo, int64_t walk_hi, int64_t tokens_total, int64_t token_offset, uint64_t seed,
                                 int32_t flags, void *stream)
{
    GEMHIP_REQUIRE(h && h->SynPos, "sgns_train: call sgns_init first");
    GEMHIP_REQUIRE(h->unigram_ready, "sgns_train: call n2v_vocab + n2v_build_unigram first");
    GEMHIP_REQUIRE(neg == SGNS_NEG, "sgns_train: neg=%d unsupported (the reference binary fixes NegSamN=5)", neg);
    GEMHIP_REQUIRE(window >= 1 && window < 16384, "sgns_train: window=%d", window);
    GEMHIP_REQUIRE(epochs >= 1 && epoch >= 0 && epoch < epochs && epoch < 256, "sgns_train: epoch %d of %d", epoch, epochs);
    GEMHIP_REQUIRE(0 <= walk_lo && walk_lo <= walk_hi && walk_hi <= h->nwalks, "sgns_train: bad local walk range");
    GEMHIP_REQUIRE(tokens_total >= 1, "sgns_train: tokens_total=%lld", (long long)tokens_total);
    if (walk_hi == walk_lo) return GEMHIP_OK;
    SgnsArgs A;
    A.walks = h->d_walks; A.walk_lo = walk_lo; A.walk_hi = walk_hi; A.walk_len = h->walk_len; A.window = window;
    A.alpha0 = alpha0; A.denom = (int64_t)epochs * tokens_total + 1;
    // the kernel computes t = token_offset + wl*walk_len + pos with wl the LOCAL walk index
    A.token_offset = token_offset; A.walk_id_offset = h->walk_id_offset; A.epoch = epoch;
    A.UT = h->d_UT; A.KT = h->d_KT; A.UK = h->d_UK; A.n = (uint32_t)h->n; A.seed = seed; A.flags = flags; A.d = h->d;
    if (h->vocab_order) {      // the binary's table layout: slots over the nodes that occur, in first-appearance order; the slot table is always consulted
        A.KT = h->d_KTslot; A.n = (uint32_t)h->n_vocab; A.flags = flags | 2; A.UT = nullptr;
    }
    A.SK = nullptr;
    A.SynPos = h->SynPos; A.SynNeg = h->SynNeg; A.pairs = h->d_pairs;
    A.dummy = nullptr; A.prof = nullptr; A.cache_radius = 0; A.nwaves = 1; A.prefetch = h->kn.prefetch; A.reload = h->kn.reload; A.counts = nullptr; A.hot_thr = 0;
    A.parts = 0; A.ctx_part = 0; A.word_part = 0; A.seg = nullptr; A.nseg = 0; A.seg_len = 0; A.scratch = nullptr;
    A.fresh = h->kn.fresh; A.neg_thr = h->kn.neg_count; A.stale_ver = nullptr; A.stale_hist = nullptr; A.n_nodes = h->n;
    SgnsKnobs kn_launch = h->kn;
    kn_launch.node_id_layout = !h->vocab_order;
    // LOCALLY HOT ROWS need the WHOLE corpus on this handle (walks-per-node is a property of the corpus: a rank's shard against all-reduced counts would
    // flag everything); Hogwild launches only
    if (!(flags & 4) && h->kn.local_hot > 0 && h->walk_id_offset == 0 && h->vs.total <= (double)h->nwalks * h->walk_len + 0.5) {
        { const int rc = ensure_hotkey(h, (hipStream_t)stream); if (rc) return rc; }
        kn_launch.has_local_hot = h->n_local_hot > 0;
    }
    const SgnsLaunchPlan P = plan_sgns_launch(h->vs, kn_launch, h->n, h->d, window, h->walk_len, walk_hi - walk_lo, flags);
    GEMHIP_REQUIRE(P.lds <= 64 * 1024, "sgns_train: walk_len/window/d too large for LDS staging (%zu bytes)", P.lds);
    A.nwaves = (int32_t)P.waves; A.cache_radius = P.R;
    if (P.window) {
        { const int rc = ensure_slot_table(h, A.KT, (int64_t)A.n, (A.flags & 2) ? 1 : 0, (hipStream_t)stream); if (rc) return rc; }
        A.SK = h->d_SK;
        // hot rows (sgns_win_kernel): never cached; the launch then takes the instantiation that handles uncached contexts
        A.counts = h->d_counts; A.hot_thr = P.hot_thr;
        // LOCALLY HOT ROWS: the kernel compares `hotkey` (INT32_MAX for a node whose tokens are packed into few walks, else its token count) with the threshold
        if (P.delta && P.waves > 1 && kn_launch.has_local_hot) A.counts = h->d_hotkey;
        const size_t need = (size_t)P.waves * sgns_win_row_floats(h->d) * sizeof(float);
        if (need > h->dummy_bytes) {
            if (h->d_dummy) { GEMHIP_CHECK(hipDeviceSynchronize()); hipFree(h->d_dummy); h->d_dummy = nullptr; h->dummy_bytes = 0; }
            GEMHIP_CHECK(hipMalloc(&h->d_dummy, need));
            GEMHIP_CHECK(hipMemset(h->d_dummy, 0, need));
            h->dummy_bytes = need;
        }
        A.dummy = h->d_dummy;
#ifdef GEMHIP_SGNS_STALENESS
        static unsigned int *d_sver = nullptr; static unsigned long long *d_shist = nullptr; static int64_t sver_n = 0;
        if (sver_n < h->n) { if (d_sver) hipFree(d_sver); GEMHIP_CHECK(hipMalloc(&d_sver, (size_t)h->n * 2 * sizeof(unsigned int))); sver_n = h->n; }
        if (!d_shist) GEMHIP_CHECK(hipMalloc(&d_shist, 3 * 32 * 16 * sizeof(unsigned long long)));
        GEMHIP_CHECK(hipMemset(d_sver, 0, (size_t)h->n * 2 * sizeof(unsigned int)));
        GEMHIP_CHECK(hipMemset(d_shist, 0, 3 * 32 * 16 * sizeof(unsigned long long)));
        A.stale_ver = d_sver; A.stale_hist = d_shist;
#endif
#ifdef GEMHIP_SGNS_PROFILE
        static unsigned long long *d_prof = nullptr;
        if (!d_prof) GEMHIP_CHECK(hipMalloc(&d_prof, 64));
        GEMHIP_CHECK(hipMemset(d_prof, 0, 64));
        A.prof = d_prof;
#endif
    }
    sgns_fn fn = !P.window ? pick_sgns(h->d) : P.delta ? pick_sgns_win_hogwild(h->d) : pick_sgns_win_det(h->d);
    GEMHIP_REQUIRE(fn != nullptr, "sgns_train: d=%d unsupported", h->d);
    h->last_plan = P; h->last_fresh = (P.window && P.delta && P.hot_thr > 0) ? A.fresh : 0;
    fn(A, P.blocks, P.threads, P.lds, (hipStream_t)stream);
    GEMHIP_CHECK(hipGetLastError());
#ifdef GEMHIP_SGNS_STALENESS
    if (P.window) {      // one JSON line per launch: {"waves", "hot_thr", "fresh", "hist": {class: {log2 count: [launches by bit length of the foreign-update count]}}}
        std::vector<unsigned long long> hh(3 * 32 * 16);
        GEMHIP_CHECK(hipDeviceSynchronize());
        GEMHIP_CHECK(hipMemcpy(hh.data(), A.stale_hist, hh.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
        FILE *fo = getenv("GEMHIP_SGNS_STALENESS_OUT") ? fopen(getenv("GEMHIP_SGNS_STALENESS_OUT"), "a") : stderr;
        if (fo) {
            fprintf(fo, "{\"waves\": %lld, \"hot_thr\": %d, \"fresh\": %d, \"vocab_order\": %d, \"hist\": {", (long long)P.waves, (int)P.hot_thr, (int)A.fresh, (int)h->vocab_order);
            const char *cls[3] = {"negative", "centre", "context"};
            for (int c = 0; c < 3; ++c) {
                fprintf(fo, "%s\"%s\": {", c ? ", " : "", cls[c]);
                bool first = true;
                for (int hb = 0; hb < 32; ++hb) {
                    unsigned long long tot = 0; for (int sb = 0; sb < 16; ++sb) tot += hh[(c * 32 + hb) * 16 + sb];
                    if (!tot) continue;
                    fprintf(fo, "%s\"%d\": [", first ? "" : ", ", hb); first = false;
                    for (int sb = 0; sb < 16; ++sb) fprintf(fo, "%s%llu", sb ? ", " : "", hh[(c * 32 + hb) * 16 + sb]);
                    fprintf(fo, "]");
                }
                fprintf(fo, "}");
            }
            fprintf(fo, "}}\n");
            if (fo != stderr) fclose(fo);
        }
    }
#endif
#ifdef GEMHIP_SGNS_PROFILE
    if (P.window) {
        unsigned long long hp[8];
        GEMHIP_CHECK(hipDeviceSynchronize());
        GEMHIP_CHECK(hipMemcpy(hp, A.prof, 64, hipMemcpyDeviceToHost));
        fprintf(stderr, "[sgns profile] waves=%lld cycles: other=%llu issue=%llu ctx_lds=%llu wait_rows=%llu compute_store=%llu neg_pipeline=%llu centre_setup=%llu centre_end=%llu\n",
                (long long)P.waves, hp[0], hp[1], hp[2], hp[3], hp[4], hp[5], hp[6], hp[7]);
    }
#endif
    return GEMHIP_OK;
}

// The launch plan for the given token counts, without a device: what gemhip_sgns_train would do on a handle with default knobs (tests/test_sgns_plan.py).
extern "C" int gemhip_sgns_plan_launch(const int32_t *counts, int64_t n, int32_t d, int32_t window, int32_t walk_len, int64_t nwalks, int32_t flags,
                                       int32_t *kernel, int32_t *waves, int32_t *hot_threshold, double *n_eff, double *n_eff_cold)
{
    GEMHIP_REQUIRE(counts && n >= 1 && d >= 1 && window >= 1 && walk_len >= 1 && nwalks >= 1, "sgns_plan_launch: bad arguments");
    VocabStats vs;
    vs.build(counts, n);
    SgnsKnobs kn;
    kn.node_id_layout = !(flags & GEMHIP_N2V_VOCAB_ORDER);
    const SgnsLaunchPlan P = plan_sgns_launch(vs, kn, n, d, window, walk_len, nwalks, flags);
    if (kernel) *kernel = !P.window ? 0 : P.delta ? 2 : 1;          // 0 sgns_kernel, 1 sgns_win_kernel (overwrite on leave), 2 sgns_win_kernel (Hogwild: delta write-back)
    if (waves) *waves = (int32_t)P.waves;
    if (hot_threshold) *hot_threshold = P.hot_thr;
    if (n_eff) *n_eff = vs.n_eff;
    if (n_eff_cold) *n_eff_cold = P.hot_thr > 0 ? vs.n_eff_cold((double)P.hot_thr) : vs.n_eff;
    return GEMHIP_OK;
}

// One bucket of the partitioned (N-GPU) schedule trained in WALK order: TrainModel over a walk corpus restricted to the pairs whose context lies in
// partition ctx_part (rows of dSynPos_part) and whose centre word lies in partition word_part (rows of dSynNeg_part; negatives drawn from the unigram
// table restricted to word_part).  The corpus may be assembled from several ranks' shards (d_seg: nseg segments, see SgnsArgs::seg); alpha follows
// the index of the token in the work-item order (token_offset + index).  Same kernel, same draws per (walk id, position) as gemhip_sgns_train: with
// parts == 1 it IS gemhip_sgns_train.  Replaces the pair-list pipeline (emit -> bucket -> all-to-all -> sgns_pairs_kernel) of rounds 2-3: what a
// rank needs from the others is their walks (4 bytes per token, once), not pairs (80 bytes per token, every episode), and the order of the pairs
// inside a bucket is the reference's.
extern "C" int gemhip_sgns_train_part(gemhip_n2v_t h, const void *d_walks, int64_t nwalks, int32_t walk_len, const void *d_seg, int32_t nseg,
                                      int64_t seg_len, int64_t walk_id_offset, int32_t window, float alpha0, int64_t alpha_tokens_total,
                                      int64_t token_offset, int32_t epoch, uint64_t seed, int32_t flags, int32_t ctx_part, int32_t word_part,
                                      void *dSynPos_part, void *dSynNeg_part, int32_t d, void *stream)
{
    GEMHIP_REQUIRE(h && h->parts >= 1, "sgns_train_part: call n2v_build_unigram_parts first");
    GEMHIP_REQUIRE(nwalks >= 0 && (nwalks == 0 || d_walks) && walk_len >= 1 && walk_len < 65536 && dSynPos_part && dSynNeg_part, "sgns_train_part: bad arguments");
    GEMHIP_REQUIRE(ctx_part >= 0 && ctx_part < h->parts && word_part >= 0 && word_part < h->parts, "sgns_train_part: partitions (%d, %d) of %d", ctx_part, word_part, h->parts);
    GEMHIP_REQUIRE(window >= 1 && window < 16384 && epoch >= 0 && epoch < 256 && alpha_tokens_total >= 1, "sgns_train_part: window=%d epoch=%d", window, epoch);
    GEMHIP_REQUIRE(d_seg == nullptr || (nseg >= 1 && seg_len >= 1 && nwalks == (int64_t)nseg * seg_len), "sgns_train_part: nseg=%d seg_len=%lld nwalks=%lld (need nwalks == nseg * seg_len work items)",
                   nseg, (long long)seg_len, (long long)nwalks);
    GEMHIP_REQUIRE((h->n + h->parts - 1) / h->parts < (int64_t)(1 << 29), "sgns_train_part: more than 2^29 rows per partition");
    if (nwalks == 0) return GEMHIP_OK;
    SgnsArgs A;
    A.walks = (const int32_t *)d_walks; A.walk_lo = 0; A.walk_hi = nwalks; A.walk_len = walk_len; A.window = window;
    A.alpha0 = alpha0; A.denom = alpha_tokens_total + 1; A.token_offset = token_offset; A.walk_id_offset = walk_id_offset; A.epoch = epoch;
    const int64_t off = h->part_off[word_part];
    A.UT = h->d_UTp + off; A.KT = h->d_KTp + off; A.UK = h->d_UKp + off; A.n = (uint32_t)(h->part_off[word_part + 1] - off);
    A.seed = seed; A.flags = flags; A.d = d;
    if (h->parts_vocab_order) {        // the binary's layout: slots over the partition's nodes that occur; the slot table is always consulted
        A.KT = h->d_KTslotp + off; A.n = (uint32_t)h->part_slots[word_part]; A.flags = flags | 2; A.UT = nullptr;
    }
    { const int rc = ensure_slot_table_parts(h, (A.flags & 2) ? 1 : 0, (hipStream_t)stream); if (rc) return rc; }
    A.SK = h->d_SKp + off;
    A.SynPos = (float *)dSynPos_part; A.SynNeg = (float *)dSynNeg_part; A.pairs = h->d_pairs;
    A.dummy = nullptr; A.prof = nullptr; A.prefetch = 2; A.reload = 1; A.counts = h->d_counts; A.hot_thr = 0;
    A.parts = h->parts; A.ctx_part = ctx_part; A.word_part = word_part; A.seg = (const int64_t *)d_seg; A.nseg = nseg; A.seg_len = seg_len;
    A.fresh = h->kn.fresh; A.neg_thr = h->kn.neg_count; A.stale_ver = nullptr; A.stale_hist = nullptr; A.n_nodes = h->n;
    // the launch rule of gemhip_sgns_train on the rows in play: the negative rows are those of partition word_part (n_eff of ITS restricted unigram
    // distribution bounds the Hogwild width: rho = W x 5 x 0.4 / n_eff <= 1.5 %); hot rows are judged on the GLOBAL token counts (a hub sits in
    // W x (2R+1) x count / tokens windows whatever partition it belongs to)
    VocabStats vs = h->vs_part[word_part];
    vs.total = h->vs.total; vs.max = h->vs.max; vs.touch2 = h->vs.touch2;
    SgnsKnobs kn = h->kn;
    kn.prefetch = 2; kn.reload = 1; kn.part = true; kn.node_id_layout = !h->parts_vocab_order;
    // Duty cycle.  The rule bounds the negative rows that are OPEN at any time (W x 5 x w of them).  A wavefront of a bucket launch spends only part of
    // its time in pair steps: per walk it has walk_len x (window + 1) x 0.95 / parts^2 pairs to train but still 2 x walk_len / parts rows (the contexts and
    // the centre words of its two partitions) to fetch and return, each an exposed round trip of ~0.7 pair steps.  With that fraction f of the time in
    // pair steps the same bound on open rows allows W / f wavefronts (f = 0.88 for one partition -- left at 1, the validated rule --, 0.65 at 4, 0.47 at 8).
    if (h->parts > 1) {
        const double pairs_pp = (double)walk_len * (window + 1) * 0.95 / ((double)h->parts * h->parts), rows_pp = 2.0 * walk_len / h->parts;
        kn.duty = pairs_pp / (pairs_pp + 0.7 * rows_pp);
        // the pairs a bucket launch trains all have their context in ONE partition and their negatives in ONE partition: a row of those partitions is touched
        // `parts` times as often per trained pair as in the whole corpus (sum over the partition's rows of (parts x load)^2 = parts x touch2), in the
        // fraction `duty` of the time
        kn.touch_scale = (double)h->parts * kn.duty;
        // from ~4 partitions on a walk's contexts of one partition fit the window's slots and stay cached for the WHOLE walk (sgns_win_kernel<PART>,
        // whole-walk mode): a node then sits in W x walk_len x count / tokens windows, which is what decides whether it is hot
        if (walk_len / h->parts <= 2 * std::min(window, 10) + 1) kn.window_span = walk_len;
    }
    kn.has_local_hot = !(flags & 4) && h->hotkey_state == 2 && h->n_local_hot > 0;       // LOCALLY HOT ROWS of the gathered corpus (gemhip_n2v_locally_hot_corpus)
    const SgnsLaunchPlan P = plan_sgns_launch(vs, kn, (int64_t)A.n, d, window, walk_len, nwalks, flags);
    GEMHIP_REQUIRE(P.window, "sgns_train_part: d=%d window=%d walk_len=%d do not fit the LDS window kernel", d, window, walk_len);
    A.nwaves = (int32_t)P.waves; A.cache_radius = P.R; A.hot_thr = P.hot_thr;
    if (kn.has_local_hot && P.delta && P.waves > 1) A.counts = h->d_hotkey;
    const size_t need = (size_t)P.waves * sgns_win_row_floats(d) * sizeof(float);
    if (need > h->dummy_bytes) {
        if (h->d_dummy) { GEMHIP_CHECK(hipDeviceSynchronize()); hipFree(h->d_dummy); h->d_dummy = nullptr; h->dummy_bytes = 0; }
        GEMHIP_CHECK(hipMalloc(&h->d_dummy, need));
        GEMHIP_CHECK(hipMemset(h->d_dummy, 0, need));
        h->dummy_bytes = need;
    }
    A.dummy = h->d_dummy;
    A.scratch = nullptr;
    if (P.delta) {
        const size_t sneed = (size_t)P.waves * (size_t)(2 * P.R + 1) * sgns_win_row_floats(d) * sizeof(float);
        if (sneed > h->scratch_bytes) {
            if (h->d_scratch) { GEMHIP_CHECK(hipDeviceSynchronize()); hipFree(h->d_scratch); h->d_scratch = nullptr; h->scratch_bytes = 0; }
            GEMHIP_CHECK(hipMalloc(&h->d_scratch, sneed));
            h->scratch_bytes = sneed;
        }
        A.scratch = h->d_scratch;
    }
    sgns_fn fn = pick_sgns_win_part(d, P.delta);
    GEMHIP_REQUIRE(fn != nullptr, "sgns_train_part: d=%d unsupported", d);
    h->last_plan = P; h->last_fresh = (P.delta && P.hot_thr > 0) ? A.fresh : 0;
    fn(A, P.blocks, P.threads, P.lds, (hipStream_t)stream);
    GEMHIP_CHECK(hipGetLastError());
    return GEMHIP_OK;
}

// D2D copy of local walks [walk_lo, walk_hi) into a caller-owned device buffer (e.g. a torch tensor that an all-gather then assembles the corpus from)
extern "C" int gemhip_n2v_copy_walks(gemhip_n2v_t h, int64_t walk_lo, int64_t walk_hi, void *d_dst, void *stream)
{
    GEMHIP_REQUIRE(h && d_dst && 0 <= walk_lo && walk_lo <= walk_hi && walk_hi <= h->nwalks, "n2v_copy_walks: bad arguments");
    if (walk_hi > walk_lo)
        GEMHIP_CHECK(hipMemcpyAsync(d_dst, h->d_walks + walk_lo * h->walk_len, (size_t)(walk_hi - walk_lo) * h->walk_len * sizeof(int32_t), hipMemcpyDeviceToDevice,
                                    (hipStream_t)stream));
    return GEMHIP_OK;
}

namespace {
__global__ void wave_sum6_test_kernel(const float *in, float *out)
{
    const int lane = lane_id();
    float p[6];
    for (int k = 0; k < 6; ++k) p[k] = in[lane * 6 + k];
    out[lane] = wave_sum6(p, lane);
}
}  // namespace

// Building block exposed for its own parity test: in[64][6] partial sums of one wavefront -> out[64], lane l receiving the
// wave total of value (l & 4) ? 4 + (l & 1) : (l & 3).
extern "C" int gemhip_test_wave_sum6(const float *in_host, float *out_host)
{
    GEMHIP_REQUIRE(in_host && out_host, "test_wave_sum6: NULL argument");
    float *d = nullptr;
    GEMHIP_CHECK(hipMalloc(&d, (64 * 6 + 64) * sizeof(float)));
    hipError_t e = hipMemcpy(d, in_host, 64 * 6 * sizeof(float), hipMemcpyHostToDevice);
    if (e == hipSuccess) { hipLaunchKernelGGL(wave_sum6_test_kernel, dim3(1), dim3(64), 0, 0, d, d + 64 * 6); e = hipGetLastError(); }
    if (e == hipSuccess) e = hipMemcpy(out_host, d + 64 * 6, 64 * sizeof(float), hipMemcpyDeviceToHost);
    hipFree(d);
    if (e != hipSuccess) return fail(GEMHIP_E_HIP, "test_wave_sum6: %s", hipGetErrorString(e));
    return GEMHIP_OK;
}

extern "C" int gemhip_sgns_set_hogwild(gemhip_n2v_t h, int32_t prefetch_pairs, int32_t reload_on_update)
{
    GEMHIP_REQUIRE(h && prefetch_pairs >= 0 && prefetch_pairs <= 2 && reload_on_update >= -1 && reload_on_update <= 1, "sgns_set_hogwild: bad arguments");
    if (prefetch_pairs) h->kn.prefetch = prefetch_pairs;
    if (reload_on_update >= 0) h->kn.reload = reload_on_update;
    return GEMHIP_OK;
}

extern "C" int gemhip_sgns_set_hot_rows(gemhip_n2v_t h, int32_t min_count)
{
    GEMHIP_REQUIRE(h && min_count >= -1, "sgns_set_hot_rows: bad arguments");
    h->kn.hot_count = min_count;
    return GEMHIP_OK;
}

// ... the same key from a walk CORPUS assembled from every rank's shard (the partitioned N-GPU schedule: gemhip_sgns_train_part trains buckets of it), with
// the handle's -- all-reduced -- token counts: rows of -1 tokens (padding of shorter shards) count nothing.  Bucket launches on this handle then treat the
// locally hot nodes as hot rows (hotkey is indexed by GLOBAL node id, like the counts the bucket kernels read).
extern "C" int gemhip_n2v_locally_hot_corpus(gemhip_n2v_t h, const void *d_corpus, int64_t corpus_rows, int32_t walk_len, int32_t per_walk, int64_t *count, void *stream)
{
    GEMHIP_REQUIRE(h && d_corpus && corpus_rows >= 1 && walk_len >= 1 && per_walk >= -1, "n2v_locally_hot_corpus: bad arguments");
    if (per_walk >= 0) h->kn.local_hot = per_walk;
    h->hotkey_state = 0; h->n_local_hot = 0;
    if (h->kn.local_hot == 0) { if (count) *count = 0; return GEMHIP_OK; }
    hipStream_t s = (hipStream_t)stream;
    if (!h->d_hotkey) {
        GEMHIP_CHECK(hipMalloc((void **)&h->d_hotkey, (size_t)h->n * sizeof(int32_t)));
        GEMHIP_CHECK(hipMalloc((void **)&h->d_wcount, (size_t)h->n * sizeof(int32_t)));
        GEMHIP_CHECK(hipMalloc((void **)&h->d_nlocal, sizeof(unsigned int)));
    }
    GEMHIP_CHECK(hipMemsetAsync(h->d_wcount, 0, (size_t)h->n * sizeof(int32_t), s));
    GEMHIP_CHECK(hipMemsetAsync(h->d_nlocal, 0, sizeof(unsigned int), s));
    hipLaunchKernelGGL(n2v_walk_presence_kernel, dim3((unsigned)std::min<int64_t>(corpus_rows, 256 * 32)), dim3(64), (size_t)walk_len * sizeof(int32_t), s, (const int32_t *)d_corpus,
                       corpus_rows, walk_len, h->d_wcount);
    hipLaunchKernelGGL(n2v_hotkey_kernel, dim3((unsigned)((h->n + 255) / 256)), dim3(256), 0, s, h->n, h->d_counts, h->d_wcount, h->kn.local_hot, h->d_hotkey, h->d_nlocal);
    GEMHIP_CHECK(hipGetLastError());
    unsigned int nl = 0;
    GEMHIP_CHECK(hipMemcpyAsync(&nl, h->d_nlocal, sizeof nl, hipMemcpyDeviceToHost, s));
    GEMHIP_CHECK(hipStreamSynchronize(s));
    h->n_local_hot = nl;
    h->hotkey_state = 2;                 // built from a corpus: gemhip_sgns_train_part uses it; gemhip_sgns_train rebuilds its own from the handle's walks
    if (count) *count = nl;
    return GEMHIP_OK;
}

extern "C" int gemhip_n2v_locally_hot(gemhip_n2v_t h, int32_t per_walk, int64_t *count, int32_t *hotkey_host)
{
    GEMHIP_REQUIRE(h && per_walk >= -1, "n2v_locally_hot: bad arguments");
    GEMHIP_REQUIRE(h->d_walks && h->nwalks > 0, "n2v_locally_hot: no walks on this handle");
    if (per_walk >= 0 && per_walk != h->kn.local_hot) { h->kn.local_hot = per_walk; h->hotkey_state = 0; }
    if (h->kn.local_hot == 0) {            // off: nothing is locally hot, the key is the token count
        if (count) *count = 0;
        if (hotkey_host) GEMHIP_CHECK(hipMemcpy(hotkey_host, h->d_counts, (size_t)h->n * sizeof(int32_t), hipMemcpyDeviceToHost));
        return GEMHIP_OK;
    }
    { const int rc = ensure_hotkey(h, nullptr); if (rc) return rc; }
    if (count) *count = h->n_local_hot;
    if (hotkey_host) GEMHIP_CHECK(hipMemcpy(hotkey_host, h->d_hotkey, (size_t)h->n * sizeof(int32_t), hipMemcpyDeviceToHost));
    return GEMHIP_OK;
}

extern "C" int gemhip_sgns_set_fresh(gemhip_n2v_t h, int32_t bits)
{
    GEMHIP_REQUIRE(h && bits >= 0 && bits <= 7, "sgns_set_fresh: bits=%d (0..7)", bits);
    h->kn.fresh = bits;
    return GEMHIP_OK;
}

extern "C" int gemhip_sgns_last_launch(gemhip_n2v_t h, int32_t *kernel, int32_t *waves, int32_t *hot_threshold, int32_t *fresh)
{
    GEMHIP_REQUIRE(h, "sgns_last_launch: null handle");
    const SgnsLaunchPlan &P = h->last_plan;
    if (kernel) *kernel = !P.window ? 0 : P.delta ? 2 : 1;
    if (waves) *waves = (int32_t)P.waves;
    if (hot_threshold) *hot_threshold = P.hot_thr;
    if (fresh) *fresh = h->last_fresh;
    return GEMHIP_OK;
}

extern "C" int gemhip_sgns_set_window_cache(gemhip_n2v_t h, int32_t radius, int32_t delta_writeback)
{
    GEMHIP_REQUIRE(h && radius >= -1 && radius <= 31 && delta_writeback >= -1 && delta_writeback <= 1, "sgns_set_window_cache: bad arguments");
    h->kn.cache_radius = radius;
    h->kn.cache_delta = delta_writeback;
    return GEMHIP_OK;
}

// One-shot drop-in for `node2vec -i -o -d -l -r -k -e -p -q -dr -w` (node2vec.py:34-53).
extern "C" int gemhip_n2v_train(int64_t n, int64_t nnz, const int64_t *row_ptr, const int32_t *col, const float *w, int32_t d,
                                int32_t walk_len, int32_t num_walks, int32_t window, int32_t epochs, float p, float q,
                                uint64_t seed, int32_t flags, float *X_out, double *stats)
{
    GEMHIP_REQUIRE(X_out != nullptr, "n2v_train: X_out is NULL");
    for (int k = 0; k < PH_COUNT; ++k) phase_acc()[k] = 0.0;
    const double t_call = phase_now();
    gemhip_n2v_t h = nullptr;
    int rc = gemhip_n2v_create(n, nnz, row_ptr, col, w, &h);
    if (rc) return rc;
    hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr};
    for (auto &e : ev) if (!rc && hipEventCreate(&e) != hipSuccess) rc = fail(GEMHIP_E_HIP, "n2v_train: hipEventCreate");
    const int64_t nwalks = h->m_start * (int64_t)num_walks;
    if (!rc) { hipEventRecord(ev[0], 0); rc = gemhip_n2v_walks(h, p, q, num_walks, walk_len, seed, flags, 0, nwalks, nullptr); }
    if (!rc) rc = gemhip_n2v_vocab(h, nullptr);
    if (!rc) { hipEventRecord(ev[1], 0); rc = (flags & GEMHIP_N2V_VOCAB_ORDER) ? gemhip_n2v_build_unigram_vocab_order(h, flags, nullptr, nullptr, nullptr, nullptr)
                                                                                 : gemhip_n2v_build_unigram(h, nullptr, nullptr, nullptr); }
    if (!rc) rc = gemhip_sgns_init(h, d, seed, nullptr, nullptr);
    if (!rc) hipEventRecord(ev[2], 0);
    for (int ep = 0; !rc && ep < epochs; ++ep)
        rc = gemhip_sgns_train(h, window, SGNS_NEG, 0.025f, epochs, ep, 0, nwalks, nwalks * walk_len, (int64_t)ep * nwalks * walk_len,
                               seed, flags, nullptr);
    if (!rc) { hipEventRecord(ev[3], 0); rc = gemhip_sgns_get_tables(h, X_out, nullptr); }
    if (!rc && stats) {
        float a = 0, b = 0;
        hipEventElapsedTime(&a, ev[0], ev[1]);
        hipEventElapsedTime(&b, ev[2], ev[3]);
        stats[0] = a * 1e-3; stats[1] = b * 1e-3; stats[2] = (double)nwalks * walk_len; stats[3] = h->uniform_rows ? 1.0 : 0.0;
    }
    if (!rc) {     // kernel seconds of the call (HIP events): walks + vocabulary, then table init + SGNS
        float a = 0, b = 0;
        hipEventElapsedTime(&a, ev[0], ev[1]);
        hipEventElapsedTime(&b, ev[2], ev[3]);
        phase_acc()[PH_KERNELS] = (a + b) * 1e-3;
    }
    for (auto &e : ev) if (e) hipEventDestroy(e);
    gemhip_n2v_destroy(h);
    phase_acc()[PH_TOTAL] = phase_now() - t_call;
    return rc;
}
