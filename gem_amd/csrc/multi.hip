// multi.hip -- the N-GPU entry points of the C ABI: ONE process drives n_gpus devices, collectives over RCCL (xGMI).
//
// the reference has no multi-device path (`grep -r nccl\|mpi /root/reference` is empty); north_star shards GF by source row and
// node2vec by start node.  A GEM maintainer who binds libgem_hip.so gets that from two calls that mirror the one-shot drop-ins and take n_gpus:
//   gemhip_gf_train_multi    gf.py:81-101 / `gf <graph> <emb> ...` -- source rows in N contiguous blocks, the table replicated, an in-place
//                            ncclAllGather of the owned row blocks after EVERY sweep: bit-identical to gemhip_gf_train (exchange after every
//                            sweep is what keeps gf.py:93-100's Gauss-Seidel order across ranks)
//   gemhip_n2v_train_multi   node2vec.py:27-54 -- walks by start-node shard, ncclAllReduce of the token counts, ONE ncclAllGather of the walk
//                            shards, then the partitioned-table schedule: rank g keeps SynPos partition g, trains the bucket
//                            (g, (g+s) % N) of each episode in walk order (gemhip_sgns_train_part) and passes its SynNeg partition around a ring
//                            (ncclSend / ncclRecv) between rounds
// gem_amd/multi_gpu.py drives the same schedules with one process per GPU over torch.distributed (what bench.py --gpus N launches); this file is the
// same thing for a host that is not Python.  RCCL is loaded with dlopen on first use (librccl.so.1), so the library itself has no link-time
// dependency on it and every single-GPU entry point works on a box without RCCL.
//
// Test mode: a device list with REPEATED entries (e.g. {0, 0, 0, 0}) runs the ranks as VIRTUAL ranks on one device -- RCCL refuses two ranks on one
// GPU, so the three collectives are then plain device-to-device copies on one stream; everything else (sharding, episode table, bucket order, ring)
// is the production code.  That is how the one-GPU test box checks N > 1 (tests/test_multi_capi_gpu.py).
#include "common.hpp"
#include <rccl/rccl.h>
#include <dlfcn.h>
#include <vector>
#include <string>
#include <algorithm>
#include <cstring>
#include <chrono>

using namespace gemhip;

namespace {

struct RcclApi {
    void *lib = nullptr;
    std::string why;                         // dlerror() text of the failed dlopen / the first missing symbol (dlerror clears itself when read)
    decltype(&ncclCommInitAll) CommInitAll = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclAllGather) AllGather = nullptr;
    decltype(&ncclAllReduce) AllReduce = nullptr;
    decltype(&ncclSend) Send = nullptr;
    decltype(&ncclRecv) Recv = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
    bool ok = false;
};

RcclApi &rccl()
{
    static RcclApi R;
    if (R.lib || R.ok) return R;
    // The RCCL that belongs to the HIP runtime THIS library is bound to: a process can hold two (PyTorch ships its own libamdhip64 / libhsa-runtime64 /
    // librccl next to /opt/rocm's), and an RCCL bound to the other, never-initialised runtime fails inside ncclCommInitAll with "no ROCm-capable device"
    // (seen in round 6 when torch was imported AFTER this library: dlopen("librccl.so.1") then returned torch's copy by soname).  dladdr on a HIP entry
    // point names the runtime in use; its directory is searched first.
    std::vector<std::string> names;
    {
        Dl_info info;
        if (dladdr((void *)&hipGetDeviceCount, &info) && info.dli_fname) {
            std::string dir(info.dli_fname);
            const size_t cut = dir.rfind('/');
            if (cut != std::string::npos) { dir.resize(cut); names.push_back(dir + "/librccl.so.1"); names.push_back(dir + "/librccl.so"); }
        }
    }
    for (const char *fallback : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) names.push_back(fallback);
    for (const std::string &name : names) {
        R.lib = dlopen(name.c_str(), RTLD_NOW | RTLD_LOCAL);
        if (R.lib) break;
        const char *e = dlerror();
        if (R.why.empty() && e) R.why = e;
    }
    if (!R.lib) return R;
    R.why.clear();
#define GEMHIP_RCCL_SYM(f)                                                                             \
    do {                                                                                               \
        R.f = (decltype(R.f))dlsym(R.lib, "nccl" #f);                                                  \
        if (!R.f && R.why.empty()) R.why = "symbol nccl" #f " missing";                                \
    } while (0)
    GEMHIP_RCCL_SYM(CommInitAll); GEMHIP_RCCL_SYM(CommDestroy); GEMHIP_RCCL_SYM(AllGather); GEMHIP_RCCL_SYM(AllReduce); GEMHIP_RCCL_SYM(Send);
    GEMHIP_RCCL_SYM(Recv); GEMHIP_RCCL_SYM(GroupStart); GEMHIP_RCCL_SYM(GroupEnd); GEMHIP_RCCL_SYM(GetErrorString);
#undef GEMHIP_RCCL_SYM
    R.ok = R.CommInitAll && R.CommDestroy && R.AllGather && R.AllReduce && R.Send && R.Recv && R.GroupStart && R.GroupEnd && R.GetErrorString;
    return R;
}

#define NCCL_TRY(x)                                                                                                     \
    do {                                                                                                                \
        ncclResult_t _r = (x);                                                                                          \
        if (_r != ncclSuccess) return fail(GEMHIP_E_HIP, "%s failed: %s (%s:%d)", #x, rccl().GetErrorString(_r), __FILE__, __LINE__); \
    } while (0)

// inside an open ncclGroupStart: the group is closed before the error is returned (a dangling group would swallow every later call of the thread)
#define NCCL_TRY_IN_GROUP(x)                                                                                            \
    do {                                                                                                                \
        ncclResult_t _r = (x);                                                                                          \
        if (_r != ncclSuccess) {                                                                                        \
            rccl().GroupEnd();                                                                                          \
            return fail(GEMHIP_E_HIP, "%s failed: %s (%s:%d)", #x, rccl().GetErrorString(_r), __FILE__, __LINE__);      \
        }                                                                                                               \
    } while (0)

__global__ void add_i32_kernel(int32_t *__restrict__ acc, const int32_t *__restrict__ x, int64_t n)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) acc[i] += x[i];
}

// The ranks of one call: devices, one stream per rank, and the three collectives the schedules need.
struct Fabric {
    int N = 0;
    bool virt = false;                       // repeated devices: virtual ranks on one stream, copies instead of RCCL
    bool always_rccl = false;                // also route the (trivial) 1-rank collectives through RCCL (gemhip_rccl_selftest)
    std::vector<int> dev;
    std::vector<hipStream_t> st;
    std::vector<ncclComm_t> comm;
    int prev_dev = 0;

    int init(int32_t n_gpus, const int32_t *devices)
    {
        GEMHIP_REQUIRE(n_gpus >= 1 && n_gpus <= 64, "n_gpus=%d (1..64)", n_gpus);
        int count = 0;
        GEMHIP_CHECK(hipGetDeviceCount(&count));
        GEMHIP_CHECK(hipGetDevice(&prev_dev));
        N = n_gpus;
        dev.resize(N);
        for (int r = 0; r < N; ++r) {
            dev[r] = devices ? devices[r] : r;
            GEMHIP_REQUIRE(dev[r] >= 0 && dev[r] < count, "rank %d: device %d of %d visible (pass a device list with repeats to run VIRTUAL ranks on one GPU)", r, dev[r], count);
        }
        std::vector<int> u(dev); std::sort(u.begin(), u.end());
        virt = std::adjacent_find(u.begin(), u.end()) != u.end();
        if (virt) GEMHIP_REQUIRE(u.front() == u.back(), "virtual ranks: all ranks must name the SAME device");
        st.assign(N, nullptr);
        if (virt) {
            GEMHIP_CHECK(hipSetDevice(dev[0]));
            hipStream_t s = nullptr;
            GEMHIP_CHECK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
            for (int r = 0; r < N; ++r) st[r] = s;
            return GEMHIP_OK;
        }
        for (int r = 0; r < N; ++r) {
            GEMHIP_CHECK(hipSetDevice(dev[r]));
            GEMHIP_CHECK(hipStreamCreateWithFlags(&st[r], hipStreamNonBlocking));
        }
        RcclApi &R = rccl();
        if (!R.ok) return fail(GEMHIP_E_UNSUPPORTED, "librccl.so.1 could not be loaded (%s): the multi-GPU entry points need RCCL", R.why.empty() ? "unknown reason" : R.why.c_str());
        comm.assign(N, nullptr);
        (void)hipGetLastError();         // a sticky error left by an earlier, unrelated launch of this process would surface inside RCCL's init as "unhandled cuda error"
        NCCL_TRY(R.CommInitAll(comm.data(), N, dev.data()));
        return GEMHIP_OK;
    }
    int use(int r) { GEMHIP_CHECK(hipSetDevice(dev[r])); return GEMHIP_OK; }
    int sync_all()
    {
        // the DEVICE, not only the rank's stream: the streams are non-blocking, i.e. not ordered with the null stream, and the set-up phases use null-stream
        // hipMemcpy (device to device) / hipMemset, which return before they complete.  With a stream-only wait here the first sweep on st[r] could run
        // while the copy Xb <- Xa of gemhip_gf_train_multi was still in flight and have its rows overwritten (seen ONCE, round 6, in the closing tier:
        // gemhip_gf_train_multi on one rank differed from gemhip_gf_train; the same hole existed for the zeroed SynNeg partitions of the node2vec driver).
        for (int r = 0; r < (virt ? 1 : N); ++r) { GEMHIP_CHECK(hipSetDevice(dev[r])); GEMHIP_CHECK(hipStreamSynchronize(st[r])); GEMHIP_CHECK(hipDeviceSynchronize()); }
        return GEMHIP_OK;
    }
    void destroy()
    {
        for (int r = 0; r < (int)comm.size(); ++r) if (comm[r]) rccl().CommDestroy(comm[r]);
        comm.clear();
        for (int r = 0; r < (virt ? std::min(N, 1) : N); ++r) if (r < (int)st.size() && st[r]) { hipSetDevice(dev[r]); hipStreamSynchronize(st[r]); hipStreamDestroy(st[r]); }
        st.clear();
        hipSetDevice(prev_dev);
    }
    // every rank's buf holds N blocks of `bytes`; block r of rank r is valid on entry, all blocks on every rank on exit
    int all_gather_inplace(const std::vector<void *> &buf, size_t bytes)
    {
        if (N == 1 && !always_rccl) return GEMHIP_OK;
        if (virt) {
            for (int r = 0; r < N; ++r)
                for (int q = 0; q < N; ++q)
                    if (q != r) GEMHIP_CHECK(hipMemcpyAsync((char *)buf[q] + (size_t)r * bytes, (const char *)buf[r] + (size_t)r * bytes, bytes, hipMemcpyDeviceToDevice, st[0]));
            return GEMHIP_OK;
        }
        RcclApi &R = rccl();
        NCCL_TRY(R.GroupStart());
        for (int r = 0; r < N; ++r) NCCL_TRY_IN_GROUP(R.AllGather((const char *)buf[r] + (size_t)r * bytes, buf[r], bytes, ncclInt8, comm[r], st[r]));
        NCCL_TRY(R.GroupEnd());
        return GEMHIP_OK;
    }
    int all_reduce_sum_i32(const std::vector<int32_t *> &buf, int64_t count)
    {
        if (N == 1 && !always_rccl) return GEMHIP_OK;
        if (virt) {
            for (int r = 1; r < N; ++r) hipLaunchKernelGGL(add_i32_kernel, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, st[0], buf[0], buf[r], count);
            for (int r = 1; r < N; ++r) GEMHIP_CHECK(hipMemcpyAsync(buf[r], buf[0], (size_t)count * 4, hipMemcpyDeviceToDevice, st[0]));
            GEMHIP_CHECK(hipGetLastError());
            return GEMHIP_OK;
        }
        RcclApi &R = rccl();
        NCCL_TRY(R.GroupStart());
        for (int r = 0; r < N; ++r) NCCL_TRY_IN_GROUP(R.AllReduce(buf[r], buf[r], (size_t)count, ncclInt32, ncclSum, comm[r], st[r]));
        NCCL_TRY(R.GroupEnd());
        return GEMHIP_OK;
    }
    // rank r sends `send[r]` to rank r-1 and receives rank r+1's into `recv[r]`
    int ring_shift(const std::vector<void *> &send, const std::vector<void *> &recv, size_t bytes)
    {
        if (N == 1 && !always_rccl) { if (send[0] != recv[0]) GEMHIP_CHECK(hipMemcpyAsync(recv[0], send[0], bytes, hipMemcpyDeviceToDevice, st[0])); return GEMHIP_OK; }
        if (virt) {
            for (int r = 0; r < N; ++r) GEMHIP_CHECK(hipMemcpyAsync(recv[r], send[(r + 1) % N], bytes, hipMemcpyDeviceToDevice, st[0]));
            return GEMHIP_OK;
        }
        RcclApi &R = rccl();
        NCCL_TRY(R.GroupStart());
        for (int r = 0; r < N; ++r) {
            NCCL_TRY_IN_GROUP(R.Send(send[r], bytes, ncclInt8, (r + N - 1) % N, comm[r], st[r]));
            NCCL_TRY_IN_GROUP(R.Recv(recv[r], bytes, ncclInt8, (r + 1) % N, comm[r], st[r]));
        }
        NCCL_TRY(R.GroupEnd());
        return GEMHIP_OK;
    }
};

double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

struct DevBuf {          // frees on the device it was allocated on
    void *p = nullptr; int dev = 0;
    ~DevBuf() { if (p) { hipSetDevice(dev); hipFree(p); } }
    int alloc(int device, size_t bytes) { dev = device; GEMHIP_CHECK(hipSetDevice(device)); GEMHIP_CHECK(hipMalloc(&p, std::max<size_t>(bytes, 16))); return GEMHIP_OK; }
};

}  // namespace

// Communicator round trip: create over n_gpus devices, run the three collectives of the schedules on known patterns (`bytes` per rank), verify every
// byte, destroy.  The GPU tier runs it with n_gpus = 1 (RCCL itself: a 1-rank all-gather is a copy) and with virtual ranks.
extern "C" int gemhip_rccl_selftest(int32_t n_gpus, const int32_t *devices, int64_t bytes, double *seconds)
{
    GEMHIP_REQUIRE(bytes >= 4 && bytes % 4 == 0 && bytes <= ((int64_t)1 << 30), "rccl_selftest: bytes=%lld (multiple of 4, <= 1 GiB)", (long long)bytes);
    GEMHIP_REQUIRE(n_gpus >= 1 && n_gpus <= 64, "rccl_selftest: n_gpus=%d (1..64)", n_gpus);
    Fabric F;
    F.always_rccl = true;                    // n_gpus = 1 exercises RCCL itself: a 1-rank all-gather / all-reduce / self send-recv
    int rc = F.init(n_gpus, devices);
    const int N = n_gpus;
    const int64_t words = bytes / 4;
    std::vector<DevBuf> G(N), A(N), S(N), Rv(N);
    std::vector<void *> g(N), s(N), rv(N);
    std::vector<int32_t *> a(N);
    const double t0 = now_s();
    for (int r = 0; r < N && !rc; ++r) {
        if (!rc) rc = G[r].alloc(F.dev[r], (size_t)bytes * N);
        if (!rc) rc = A[r].alloc(F.dev[r], (size_t)bytes);
        if (!rc) rc = S[r].alloc(F.dev[r], (size_t)bytes);
        if (!rc) rc = Rv[r].alloc(F.dev[r], (size_t)bytes);
        if (rc) break;
        g[r] = G[r].p; a[r] = (int32_t *)A[r].p; s[r] = S[r].p; rv[r] = Rv[r].p;
        std::vector<int32_t> h((size_t)words * N, -1), hb((size_t)words);
        for (int64_t i = 0; i < words; ++i) { h[(size_t)r * words + i] = (int32_t)(1000003 * r + i); hb[i] = (int32_t)((r + 1) * (i % 97 + 1)); }
        if (hipMemcpy(g[r], h.data(), (size_t)bytes * N, hipMemcpyHostToDevice) != hipSuccess || hipMemcpy(a[r], hb.data(), bytes, hipMemcpyHostToDevice) != hipSuccess ||
            hipMemcpy(s[r], hb.data(), bytes, hipMemcpyHostToDevice) != hipSuccess)
            rc = fail(GEMHIP_E_HIP, "rccl_selftest: upload failed");
    }
    if (!rc) rc = F.all_gather_inplace(g, (size_t)bytes);
    if (!rc) rc = F.all_reduce_sum_i32(a, words);
    if (!rc) rc = F.ring_shift(s, rv, (size_t)bytes);
    if (!rc) rc = F.sync_all();
    for (int r = 0; r < N && !rc; ++r) {
        std::vector<int32_t> h((size_t)words * N), ha((size_t)words), hr((size_t)words);
        hipSetDevice(F.dev[r]);
        if (hipMemcpy(h.data(), g[r], (size_t)bytes * N, hipMemcpyDeviceToHost) != hipSuccess || hipMemcpy(ha.data(), a[r], bytes, hipMemcpyDeviceToHost) != hipSuccess ||
            hipMemcpy(hr.data(), rv[r], bytes, hipMemcpyDeviceToHost) != hipSuccess) { rc = fail(GEMHIP_E_HIP, "rccl_selftest: download failed"); break; }
        for (int q = 0; q < N && !rc; ++q)
            for (int64_t i = 0; i < words; ++i)
                if (h[(size_t)q * words + i] != (int32_t)(1000003 * q + i)) { rc = fail(GEMHIP_E_HIP, "rccl_selftest: all-gather: rank %d block %d word %lld wrong", r, q, (long long)i); break; }
        const int64_t tri = (int64_t)N * (N + 1) / 2;
        for (int64_t i = 0; i < words && !rc; ++i) {
            if (ha[i] != (int32_t)(tri * (i % 97 + 1))) rc = fail(GEMHIP_E_HIP, "rccl_selftest: all-reduce: rank %d word %lld wrong", r, (long long)i);
            else if (hr[i] != (int32_t)(((r + 1) % N + 1) * (i % 97 + 1))) rc = fail(GEMHIP_E_HIP, "rccl_selftest: ring shift: rank %d word %lld wrong", r, (long long)i);
        }
    }
    if (seconds) *seconds = now_s() - t0;
    F.destroy();
    return rc;
}

// gf.py:81-101 on n_gpus devices.  stats (optional, 8 doubles): {seconds of the sweeps incl. exchanges (wall, all ranks synchronised), updates per
// sweep (all ranks), rows per sweep, exchange bytes per rank per sweep, n_gpus, virtual ranks (0/1), 0, 0}.
extern "C" int gemhip_gf_train_multi(int64_t n, int64_t m, const int32_t *src, const int32_t *dst, const float *w, int32_t d, float eta, float regu,
                                     int32_t max_iter, int32_t n_gpus, const int32_t *devices, float *X_inout, double *stats)
{
    GEMHIP_REQUIRE(X_inout != nullptr && max_iter >= 0 && n > 0 && m >= 0 && d >= 1, "gf_train_multi: bad arguments");
    GEMHIP_REQUIRE(n_gpus >= 1 && n_gpus <= 64, "gf_train_multi: n_gpus=%d (1..64)", n_gpus);          // before anything is sized or divided by it
    if (n_gpus > 1) {
        // ranks read each other's rows from the PREVIOUS sweep's table: that is the reference's sweep only when no firing edge reads a row the
        // reference has already updated in the same sweep, i.e. when the firing sources are first visited in ascending id order (gf.py:93-100 over
        // graph.edges() of a graph whose nodes were inserted in id order -- every call site of the reference); other orders need one GPU
        int64_t last = -1;
        std::vector<char> seen((size_t)n, 0);
        for (int64_t e = 0; e < m; ++e) {
            GEMHIP_REQUIRE(src[e] >= 0 && src[e] < n && dst[e] >= 0 && dst[e] < n, "gf_train_multi: edge %lld out of range", (long long)e);
            if (dst[e] > src[e] && !seen[src[e]]) {
                seen[src[e]] = 1;
                if (src[e] < last) return fail(GEMHIP_E_UNSUPPORTED, "gf_train_multi: firing sources are not first visited in ascending id order (row %d after row %lld): "
                                                                     "sharding by source row would change the reference's Gauss-Seidel order; use n_gpus = 1", src[e], (long long)last);
                last = src[e];
            }
        }
    }
    Fabric F;
    int rc = F.init(n_gpus, devices);
    const int N = n_gpus;
    const int64_t block = (n + N - 1) / N, n_pad = block * N;
    std::vector<gemhip_gf_plan_t> plan(N, nullptr);
    std::vector<DevBuf> Xa(N), Xb(N);
    std::vector<float> Xpad;
    if (!rc && n_pad != n) { Xpad.assign((size_t)n_pad * d, 0.f); std::memcpy(Xpad.data(), X_inout, (size_t)n * d * sizeof(float)); }
    const float *Xsrc = n_pad != n ? Xpad.data() : X_inout;
    double upd = 0, rows = 0;
    for (int r = 0; r < N && !rc; ++r) {
        rc = F.use(r);
        const int64_t r0 = std::min<int64_t>(r * block, n), r1 = std::min<int64_t>((r + 1) * block, n);
        if (!rc) rc = gemhip_gf_plan_create(n, m, src, dst, w, d, r0, r1, &plan[r]);
        if (!rc) rc = Xa[r].alloc(F.dev[r], (size_t)n_pad * d * sizeof(float));
        if (!rc) rc = Xb[r].alloc(F.dev[r], (size_t)n_pad * d * sizeof(float));
        if (!rc && (hipMemcpy(Xa[r].p, Xsrc, (size_t)n_pad * d * sizeof(float), hipMemcpyHostToDevice) != hipSuccess ||
                    hipMemcpy(Xb[r].p, Xa[r].p, (size_t)n_pad * d * sizeof(float), hipMemcpyDeviceToDevice) != hipSuccess))
            rc = fail(GEMHIP_E_HIP, "gf_train_multi: table upload failed on rank %d", r);
        if (!rc) rc = gemhip_gf_plan_bind(plan[r], Xa[r].p, Xb[r].p);
        int64_t info[8];
        if (!rc) { rc = gemhip_gf_plan_info(plan[r], info); upd += (double)info[0]; rows += (double)info[1]; }
    }
    if (!rc) rc = F.sync_all();
    const double t0 = now_s();
    std::vector<void *> cur(N);
    for (int it = 0; it < max_iter && !rc; ++it) {
        for (int r = 0; r < N && !rc; ++r) {
            rc = F.use(r);
            if (!rc) rc = gemhip_gf_plan_sweeps(plan[r], 1, eta, regu, F.st[r]);
            if (!rc) rc = gemhip_gf_plan_current(plan[r], &cur[r]);
        }
        if (!rc) rc = F.all_gather_inplace(cur, (size_t)block * d * sizeof(float));          // owned row blocks -> every rank holds the whole new table
    }
    if (!rc) rc = F.sync_all();
    const double el = now_s() - t0;
    if (!rc) {
        void *p0 = nullptr;
        rc = F.use(0);
        if (!rc) rc = gemhip_gf_plan_current(plan[0], &p0);
        if (!rc && hipMemcpy(X_inout, p0, (size_t)n * d * sizeof(float), hipMemcpyDeviceToHost) != hipSuccess) rc = fail(GEMHIP_E_HIP, "gf_train_multi: download failed");
    }
    if (!rc && stats) {
        stats[0] = el; stats[1] = upd; stats[2] = rows; stats[3] = N > 1 ? (double)(N - 1) * block * d * 4.0 : 0.0; stats[4] = N; stats[5] = F.virt ? 1.0 : 0.0;
        stats[6] = 0.0; stats[7] = 0.0;
    }
    for (int r = 0; r < N; ++r) if (plan[r]) { hipSetDevice(F.dev[r]); gemhip_gf_plan_destroy(plan[r]); }
    F.destroy();
    return rc;
}

// node2vec.py:27-54 on n_gpus devices (module comment).  `episodes` slices of every rank's walk shard (64 = the default of gem_amd/multi_gpu.py; the
// SynNeg partitions complete one ring tour per episode).  stats (optional, 8 doubles): {walk + vocabulary + gather seconds, training seconds (wall, all
// ranks synchronised), tokens, pairs trained (all ranks), ring-shift bytes per rank per round, n_gpus, virtual ranks (0/1), bucket launches per rank}.
extern "C" int gemhip_n2v_train_multi(int64_t n, int64_t nnz, const int64_t *row_ptr, const int32_t *col, const float *w, int32_t d, int32_t walk_len,
                                      int32_t num_walks, int32_t window, int32_t epochs, float p, float q, uint64_t seed, int32_t flags, int32_t n_gpus,
                                      const int32_t *devices, int32_t episodes, float *X_out, double *stats)
{
    GEMHIP_REQUIRE(X_out != nullptr && episodes >= 1 && epochs >= 1 && epochs < 256, "n2v_train_multi: bad arguments (episodes=%d epochs=%d)", episodes, epochs);
    GEMHIP_REQUIRE(n_gpus >= 1 && n_gpus <= 64, "n2v_train_multi: n_gpus=%d (1..64)", n_gpus);         // before anything is sized or divided by it
    Fabric F;
    int rc = F.init(n_gpus, devices);
    const int N = n_gpus;
    std::vector<gemhip_n2v_t> h(N, nullptr);
    for (int r = 0; r < N && !rc; ++r) { rc = F.use(r); if (!rc) rc = gemhip_n2v_create(n, nnz, row_ptr, col, w, &h[r]); }
    int64_t m_start = 0;
    if (!rc) rc = gemhip_n2v_start_nodes(h[0], &m_start);
    const int64_t total = m_start * (int64_t)num_walks;
    auto shard = [&](int r, int64_t &lo, int64_t &hi) { lo = total * r / N; hi = total * (r + 1) / N; };
    const double t0 = now_s();
    // ---- walks of my start-node shard, token counts summed over the ranks
    std::vector<int32_t *> cnt(N, nullptr);
    for (int r = 0; r < N && !rc; ++r) {
        int64_t lo, hi; shard(r, lo, hi);
        rc = F.use(r);
        if (!rc) rc = gemhip_n2v_walks(h[r], p, q, num_walks, walk_len, seed, flags, lo, hi, F.st[r]);
        if (!rc) rc = gemhip_n2v_vocab(h[r], F.st[r]);
        void *c = nullptr;
        if (!rc) rc = gemhip_n2v_counts_ptr(h[r], &c);
        cnt[r] = (int32_t *)c;
    }
    if (!rc) rc = F.all_reduce_sum_i32(cnt, n);
    if (!rc) rc = F.sync_all();
    // ---- the walk corpus on every rank: shard r occupies rows [r * shard_rows, ...) (shorter shards padded with -1 tokens)
    int64_t shard_rows = 1;
    for (int r = 0; r < N; ++r) { int64_t lo, hi; shard(r, lo, hi); shard_rows = std::max(shard_rows, hi - lo); }
    const size_t shard_bytes = (size_t)shard_rows * walk_len * sizeof(int32_t);
    std::vector<DevBuf> corpus(N), segd(N), Pp(N), Na(N), Nb(N);
    std::vector<void *> cptr(N);
    for (int r = 0; r < N && !rc; ++r) {
        int64_t lo, hi; shard(r, lo, hi);
        rc = corpus[r].alloc(F.dev[r], shard_bytes * N);
        if (!rc && hipMemsetAsync((char *)corpus[r].p + (size_t)r * shard_bytes, 0xff, shard_bytes, F.st[r]) != hipSuccess) rc = fail(GEMHIP_E_HIP, "n2v_train_multi: memset");
        if (!rc) rc = gemhip_n2v_copy_walks(h[r], 0, hi - lo, (char *)corpus[r].p + (size_t)r * shard_bytes, F.st[r]);
        cptr[r] = corpus[r].p;
    }
    if (!rc) rc = F.all_gather_inplace(cptr, shard_bytes);
    // ---- the per-partition unigram tables, in the layout the flags ask for: node-id order, or (GEMHIP_N2V_VOCAB_ORDER, the plugin default on one GPU)
    // the binary's -- each partition's nodes in order of first appearance in the WHOLE corpus, which every rank now holds
    if (!rc) rc = F.sync_all();
    for (int r = 0; r < N && !rc; ++r) {
        rc = F.use(r);
        if (!rc) rc = (flags & GEMHIP_N2V_VOCAB_ORDER) ? gemhip_n2v_build_unigram_parts_vocab_order(h[r], N, flags, corpus[r].p, (int64_t)N * shard_rows * walk_len, nullptr, nullptr, nullptr, nullptr)
                                                       : gemhip_n2v_build_unigram_parts(h[r], N, nullptr, nullptr);
    }
    // ---- LOCALLY HOT ROWS of the gathered corpus (nodes whose tokens are packed into few walks: the ends of isolated edges): never cached by the bucket launches
    for (int r = 0; r < N && !rc; ++r) {
        rc = F.use(r);
        if (!rc) rc = gemhip_n2v_locally_hot_corpus(h[r], corpus[r].p, (int64_t)N * shard_rows, walk_len, -1, nullptr, F.st[r]);
    }
    // ---- episode table [episodes][3][N]: first row, walks present, first global walk id of every shard's slice; work items per shard = longest slice
    std::vector<int64_t> tab((size_t)episodes * 3 * N), seg_len(episodes, 1);
    for (int e = 0; e < episodes; ++e)
        for (int r = 0; r < N; ++r) {
            int64_t lo, hi; shard(r, lo, hi);
            const int64_t a = (hi - lo) * e / episodes, z = (hi - lo) * (e + 1) / episodes;
            int64_t *t = tab.data() + (size_t)e * 3 * N;
            t[r] = r * shard_rows + a; t[N + r] = z - a; t[2 * N + r] = lo + a;
            seg_len[e] = std::max(seg_len[e], z - a);
        }
    // ---- partition r of the initial tables a single GPU would draw (InitPosEmb / InitNegEmb): SynPos rows r, r + N, ...; SynNeg = 0
    const int64_t prow = (n + N - 1) / N;
    const size_t part_bytes = (size_t)prow * d * sizeof(float);
    for (int r = 0; r < N && !rc; ++r) {
        rc = F.use(r);
        if (!rc) rc = segd[r].alloc(F.dev[r], tab.size() * sizeof(int64_t));
        if (!rc && hipMemcpy(segd[r].p, tab.data(), tab.size() * sizeof(int64_t), hipMemcpyHostToDevice) != hipSuccess) rc = fail(GEMHIP_E_HIP, "n2v_train_multi: table upload");
        if (!rc && r == 0) rc = gemhip_sgns_init(h[0], d, seed, nullptr, nullptr);       // InitPosEmb / InitNegEmb of the whole table, once (rank 0's device)
        if (!rc) rc = Pp[r].alloc(F.dev[r], part_bytes);
        if (!rc) rc = Na[r].alloc(F.dev[r], part_bytes);
        if (!rc) rc = Nb[r].alloc(F.dev[r], part_bytes);
    }
    // the partitions are cut out of a host copy of that table (the handle keeps its tables private; get_tables synchronises the device)
    if (!rc) {
        std::vector<float> full((size_t)n * d), part((size_t)prow * d);
        rc = F.use(0);
        if (!rc) rc = gemhip_sgns_get_tables(h[0], full.data(), nullptr);
        for (int r = 0; r < N && !rc; ++r) {
            std::fill(part.begin(), part.end(), 0.f);
            for (int64_t l = 0; l * N + r < n; ++l) std::memcpy(&part[(size_t)l * d], &full[(size_t)(l * N + r) * d], (size_t)d * sizeof(float));
            hipSetDevice(F.dev[r]);
            if (hipMemcpy(Pp[r].p, part.data(), part_bytes, hipMemcpyHostToDevice) != hipSuccess || hipMemset(Na[r].p, 0, part_bytes) != hipSuccess ||
                hipMemset(Nb[r].p, 0, part_bytes) != hipSuccess)
                rc = fail(GEMHIP_E_HIP, "n2v_train_multi: partition upload failed on rank %d", r);
        }
    }
    if (!rc) rc = F.sync_all();
    const double t1 = now_s();
    // ---- episodes x rounds: bucket (r, (r + s) % N) on rank r, then the SynNeg partitions move one step around the ring
    std::vector<void *> ncur(N), ntmp(N);
    for (int r = 0; r < N; ++r) { ncur[r] = Na[r].p; ntmp[r] = Nb[r].p; }
    int64_t alpha_total = 0;
    for (int e = 0; e < episodes; ++e) alpha_total += seg_len[e] * N * walk_len;
    alpha_total *= epochs;
    int64_t done = 0, launches = 0;
    for (int ep = 0; ep < epochs && !rc; ++ep)
        for (int e = 0; e < episodes && !rc; ++e) {
            for (int s = 0; s < N && !rc; ++s) {
                for (int r = 0; r < N && !rc; ++r) {
                    rc = F.use(r);
                    if (!rc) rc = gemhip_sgns_train_part(h[r], corpus[r].p, (int64_t)N * seg_len[e], walk_len, (const int64_t *)segd[r].p + (size_t)e * 3 * N, N, seg_len[e], 0, window,
                                                         0.025f, alpha_total, done, ep, seed, flags, r, (r + s) % N, Pp[r].p, ncur[r], d, F.st[r]);
                    ++launches;
                }
                if (!rc && N > 1) { rc = F.ring_shift(ncur, ntmp, part_bytes); std::swap(ncur, ntmp); }
            }
            done += seg_len[e] * N * walk_len;
        }
    if (!rc) rc = F.sync_all();
    const double t2 = now_s();
    // ---- row v of the result = partition v % N, local row v / N
    double pairs = 0;
    if (!rc) {
        std::vector<float> part((size_t)prow * d);
        for (int r = 0; r < N && !rc; ++r) {
            hipSetDevice(F.dev[r]);
            if (hipMemcpy(part.data(), Pp[r].p, part_bytes, hipMemcpyDeviceToHost) != hipSuccess) { rc = fail(GEMHIP_E_HIP, "n2v_train_multi: download failed on rank %d", r); break; }
            for (int64_t l = 0; l * N + r < n; ++l) std::memcpy(X_out + (size_t)(l * N + r) * d, &part[(size_t)l * d], (size_t)d * sizeof(float));
            int64_t pr = 0;
            rc = gemhip_sgns_pairs(h[r], &pr, 1);
            pairs += (double)pr;
        }
    }
    if (!rc && stats) {
        stats[0] = t1 - t0; stats[1] = t2 - t1; stats[2] = (double)total * walk_len; stats[3] = pairs; stats[4] = N > 1 ? (double)part_bytes : 0.0; stats[5] = N;
        stats[6] = F.virt ? 1.0 : 0.0; stats[7] = (double)launches / N;
    }
    for (int r = 0; r < N; ++r) if (h[r]) { hipSetDevice(F.dev[r]); gemhip_n2v_destroy(h[r]); }
    F.destroy();
    return rc;
}
