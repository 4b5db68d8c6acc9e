/*
 * gem_hip.h -- C ABI of libgem_hip.so, the MI355X (gfx950) backend for the
 * learn_embedding() hot path of GEM's HOPE / GraphFactorization / node2vec.
 *
 * Plain C: pointers, sizes and opaque handles only.  No torch types.
 * Conventions
 *   - every function returns 0 on success, <0 on error (GEMHIP_E_*); the message
 *     is available from gemhip_last_error() (thread-local, valid until the next
 *     failing call on that thread);
 *   - the CALLER owns every host buffer; the library keeps no host pointer after
 *     a call returns.  Device state lives in opaque handles that the caller
 *     destroys;
 *   - `stream` arguments are a hipStream_t passed as void* (NULL = the null
 *     stream).  Calls taking a stream only ENQUEUE work; everything else is
 *     blocking;
 *   - row-major everywhere; embeddings are float32 [n][d] on the device
 *     (the Python layer returns float64 like the reference does).
 *
 * Each entry point names the reference interface it replaces (paths relative
 * to the GEM repository).  INTEGRATION.md shows the ctypes stub a GEM maintainer
 * would add.
 */
#ifndef GEM_HIP_H
#define GEM_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GEMHIP_VERSION 100 /* 0.1.0 */

#define GEMHIP_OK 0
#define GEMHIP_E_INVALID (-1)  /* bad argument */
#define GEMHIP_E_HIP (-2)      /* HIP runtime error (no device, OOM, launch failure) */
#define GEMHIP_E_UNSUPPORTED (-3)
#define GEMHIP_E_NOTCONVERGED (-4)

/* ------------------------------------------------------------------ runtime */
int gemhip_version(void);
const char *gemhip_last_error(void);
int gemhip_device_count(int *n);
int gemhip_set_device(int device);
/* device-side helpers so a host (numpy / torch) can hand over or fetch buffers */
int gemhip_malloc(void **dptr, int64_t bytes);
int gemhip_free(void *dptr);
int gemhip_memcpy_h2d(void *dst_dev, const void *src_host, int64_t bytes);
int gemhip_memcpy_d2h(void *dst_host, const void *src_dev, int64_t bytes);
int gemhip_synchronize(void *stream);

/* ------------------------------------------------- Graph Factorization (GF)
 * Replaces: gem/embedding/gf.py:81-101 (GraphFactorization.learn_embedding,
 * hot loop :93-100) and the native executable it shells out to,
 * gem/c_src/gf.cpp:130-169 (`gf <graph> <emb> <verbose> <weighted> <d> <eta>
 * <regu> <max_iter> <print_step>`, gf.py:54-72).
 *
 * The edge list is the wire format of that executable (gf.cpp:54-92 /
 * gem/utils/graph_util.py:129-134): m triples (src, dst, w) in the order
 * graph.edges() yields them.  Semantics are the reference's: sequential
 * (Gauss-Seidel) sweeps, only edges with dst > src update, only row `src` is
 * written.  The device schedule reproduces that order exactly (level-scheduled
 * rows + double-buffered table) -- results differ from the fp32
 * CPU loop only by dot-product summation order.
 */
typedef struct gemhip_gf_plan *gemhip_gf_plan_t;

/* One-shot drop-in for `gf` / GraphFactorization.learn_embedding.
 * X_inout: [n][d] float32, in = initial embedding (gf.cpp:41-52 draws
 * 0.01*N(0,1)), out = trained embedding.  stats (optional, 4 doubles):
 * {kernel_seconds, updates_per_sweep, rows_per_sweep, levels}. */
int gemhip_gf_train(int64_t n, int64_t m, const int32_t *src, const int32_t *dst,
                    const float *w /* NULL = all 1.0 */, int32_t d, float eta,
                    float regu, int32_t max_iter, float *X_inout, double *stats);

/* Staged form (graph resident in HBM across calls; used by bench.py and the
 * multi-GPU driver).  [row_begin,row_end) restricts the plan to the source
 * rows this rank owns (source-node sharding); pass 0,n for all. */
int gemhip_gf_plan_create(int64_t n, int64_t m, const int32_t *src,
                          const int32_t *dst, const float *w, int32_t d,
                          int64_t row_begin, int64_t row_end,
                          gemhip_gf_plan_t *out);
int gemhip_gf_plan_destroy(gemhip_gf_plan_t plan);
/* Use caller-provided DEVICE buffers [n][d] float32 for the two table copies
 * (e.g. torch tensors, so RCCL can all-gather them).  Both must hold the same
 * initial embedding.  Without this call the plan allocates its own. */
int gemhip_gf_plan_bind(gemhip_gf_plan_t plan, void *dX_a, void *dX_b);
int gemhip_gf_plan_set_embedding(gemhip_gf_plan_t plan, const float *X_host);
/* 0.01*N(0,1)-style init on the device (Philox4x32-10 + Box-Muller), same
 * distribution as gf.cpp:41-52 / gf.py:92. */
int gemhip_gf_plan_init_embedding(gemhip_gf_plan_t plan, uint64_t seed, float scale);
/* Enqueue `nsweeps` sweeps (one sweep = gf.cpp:152-164 over all edges). */
int gemhip_gf_plan_sweeps(gemhip_gf_plan_t plan, int32_t nsweeps, float eta,
                          float regu, void *stream);
int gemhip_gf_plan_get_embedding(gemhip_gf_plan_t plan, float *X_host);
/* Device pointer of the CURRENT table (the one holding the latest sweep). */
int gemhip_gf_plan_current(gemhip_gf_plan_t plan, void **dX);
/* info (8 int64): {updates_per_sweep, rows_per_sweep, levels, n, d,
 * algorithmic_bytes_per_sweep, 0, 0} */
int gemhip_gf_plan_info(gemhip_gf_plan_t plan, int64_t *info);
/* gf.cpp:94-113 objective on the device table: out = {f1, f2}. `m` edges as in
 * plan_create but over ALL edges (no dst>src filter), like the reference. */
int gemhip_gf_objective(int64_t n, int64_t m, const int32_t *src,
                        const int32_t *dst, const float *w, int32_t d,
                        const float *X_host, double *out);

#ifdef __cplusplus
}
#endif
#endif /* GEM_HIP_H */
