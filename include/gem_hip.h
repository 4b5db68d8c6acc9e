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
/* Where the wall time of the LAST one-shot call on this thread went (gemhip_gf_train, gemhip_n2v_train, gemhip_hope: the drop-ins
 * for gf.py:55-72, node2vec.py:35-48 and hope.py:28-36, whose callers time `learn_embedding` as a whole -- SURVEY 8d "API wall"):
 * out[8] = {total_seconds, host_prepare_seconds (CSR sort, alias / unigram tables, plans), h2d_seconds, kernel_seconds (HIP events),
 * d2h_seconds, 0, 0, 0}. */
int gemhip_last_call_phases(double *out);

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
 *
 * PRECONDITION (checked; GEMHIP_E_INVALID otherwise): every firing edge
 * (dst > src) reads a row that, in file order, has had either all or none of
 * its updates of the sweep -- always true when a source's edges are contiguous
 * (what graph.edges() and saveGraphToEdgeListTxt produce).  gf.cpp:152-164
 * processes an arbitrary file strictly in file order; a list such as
 * (1,2),(0,1),(1,3), where row 0 must see row 1 between its two updates, needs
 * a third version of a row and is rejected rather than silently regrouped.
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
/* Source rows a wavefront trains back to back inside one sweep launch (1 = one row per wavefront, gf_sweep_kernel; K > 1 =
 * gf_sweep_rows_kernel, which has the next row's inputs in flight while a row is trained); 0 = auto (by level size).  Every
 * setting gives bit-identical tables -- rows of a level are independent.  No reference counterpart (gf.cpp is one thread). */
int gemhip_gf_plan_set_rows_per_wave(gemhip_gf_plan_t plan, int32_t rows_per_wave);
/* Sweeps per COOPERATIVE launch (gf_sweeps_coop_kernel: a resident grid, sweeps separated by a grid barrier instead of a kernel boundary): 0 = off
 * (default: one launch per sweep and level), k > 1 = up to k sweeps per launch on single-level plans without hub rows (others keep the launch loop);
 * max_grid caps the workgroups of that launch (0 = what the occupancy allows).  Bit-identical tables either way.  No reference counterpart. */
int gemhip_gf_plan_set_fused_sweeps(gemhip_gf_plan_t plan, int32_t sweeps_per_launch, int32_t max_grid);
int gemhip_gf_plan_get_embedding(gemhip_gf_plan_t plan, float *X_host);
/* Device pointer of the CURRENT table (the one holding the latest sweep). */
int gemhip_gf_plan_current(gemhip_gf_plan_t plan, void **dX);
/* info (8 int64): {updates_per_sweep, rows_per_sweep, levels, n, d,
 * algorithmic_bytes_per_sweep, rows per wavefront of the largest level's launch, 0} */
int gemhip_gf_plan_info(gemhip_gf_plan_t plan, int64_t *info);
/* gf.cpp:94-113 objective on the device table: out = {f1, f2}. `m` edges as in
 * plan_create but over ALL edges (no dst>src filter), like the reference. */
int gemhip_gf_objective(int64_t n, int64_t m, const int32_t *src,
                        const int32_t *dst, const float *w, int32_t d,
                        const float *X_host, double *out);

/* ------------------------------------------------------------------ node2vec
 * Replaces: gem/embedding/node2vec.py:27-54 (node2vec.learn_embedding), i.e. the
 * subprocess call of the prebuilt SNAP binary gem/c_exe/node2vec with
 * `-i:tempGraph.graph -o:tempGraph.emb -d -l -r -k -e -p -q -v -dr -w`
 * (node2vec.py:35-46) and the text files around it (graph_util.py:137-140, 161-169).
 *
 * Graph: CSR by source (row_ptr int64[n+1], col int32[nnz], w float32[nnz] or NULL
 * for unit weights) -- what the binary builds from the "%d %d %f" edge list.
 *
 * flags (bit set): 1 = short walks are padded with token 0 and trained on, like the
 * binary's zero-initialised walk matrix (else padded with -1 and skipped);
 * 2 = negative sampling indexes the alias table the way the binary's RndUnigramInt
 * does (KTable[floor(u*n)], see oracle/n2v_oracle.c); 4 = deterministic: SGNS runs
 * on ONE wavefront in walk order (bit-reproducible, for parity tests; slow);
 * 8 = first hop of a walk is uniform over neighbours, like SimulateWalk.
 * GEMHIP_N2V_SNAP_COMPAT = 1|2|8 mirrors the reference binary. */
#define GEMHIP_N2V_PAD_ZERO 1
#define GEMHIP_N2V_UNIGRAM_QUIRK 2
#define GEMHIP_N2V_DETERMINISTIC 4
#define GEMHIP_N2V_UNIFORM_FIRST_HOP 8
#define GEMHIP_N2V_SNAP_COMPAT 11
#define GEMHIP_N2V_VOCAB_ORDER 16     /* unigram alias table laid out the way the binary lays it out: over the nodes that occur, in order of first appearance
                                         in the walk matrix (LearnVocab renames the tokens that way), instead of over all nodes in id order -- same distribution,
                                         the binary's TABLE (gemhip_n2v_build_unigram_vocab_order; single-GPU path).  GEMHIP_N2V_SNAP_COMPAT | 16 = 27 is what
                                         gem_amd.embedding.node2vec passes by default */
#define GEMHIP_N2V_SNAP_LAYOUT 27
#define GEMHIP_N2V_NO_WINDOW_CACHE 128 /* A/B switch: train with the round-1 kernel (every context row goes to memory for every pair) instead of the
                                          LDS-window kernel (gemhip_sgns_set_window_cache); same arithmetic and draws either way */
/* (bits 32 and 64 selected two round-1/2 experiments -- 16-byte row accesses, negatives shared per centre word -- that were measured slower /
 * are not the reference's sampling; round 3 removed them from the library, see DESIGN.md 3.3 and the git history of gem_amd/csrc/n2v.hip) */

typedef struct gemhip_n2v *gemhip_n2v_t;

/* One-shot drop-in for the binary: X_out [n][d] float32 = SynPos in node-id order
 * (what loadEmbedding reads back).  alpha0 = 0.025, 5 negatives as in the binary.
 * stats (optional, 4 doubles): {walk_seconds, sgns_seconds, tokens, rows_uniform}. */
int gemhip_n2v_train(int64_t n, int64_t nnz, const int64_t *row_ptr, const int32_t *col,
                     const float *w, int32_t d, int32_t walk_len, int32_t num_walks,
                     int32_t window, int32_t epochs, float p, float q, uint64_t seed,
                     int32_t flags, float *X_out, double *stats);

/* Staged form (graph, walks and tables resident in HBM; used by bench.py, the
 * multi-GPU driver and the parity tests). */
int gemhip_n2v_create(int64_t n, int64_t nnz, const int64_t *row_ptr, const int32_t *col,
                      const float *w, gemhip_n2v_t *out);
int gemhip_n2v_destroy(gemhip_n2v_t h);
/* PreprocessTransitionProbs: first-order Vose alias tables per row (no-op when every
 * row has equal weights).  2nd-order bias is applied by rejection inside the walk.
 * Rows of fewer than 2048 neighbours: GetNodeAlias's loop on one lane, fp32, bit for bit
 * oracle_alias_build_f32; longer (hub) rows: the loop's closed form on a workgroup, fp64
 * sums in a fixed order, bit for bit oracle_alias_build_hub (same alias targets as the
 * sequential loop). */
int gemhip_n2v_build_alias(gemhip_n2v_t h, void *stream);
/* InitUnigramTable in the binary's layout (flags bit GEMHIP_N2V_VOCAB_ORDER of the one-shot call): see n2v.hip.  Needs the walks and the counts
 * (gemhip_n2v_vocab) of the whole corpus on this handle.  `flags`: bit 2 (the RndUnigramInt quirk) decides what a slot maps to.  n_vocab_out: nodes
 * that occur; order_out[n_vocab]: their ids in first-appearance order; UT_out / KT_out [n_vocab]: the alias table in the binary's renamed indices
 * (all optional, caller-sized n).  gemhip_sgns_train uses this table from then on (gemhip_n2v_build_unigram switches back). */
int gemhip_n2v_build_unigram_vocab_order(gemhip_n2v_t h, int32_t flags, int64_t *n_vocab_out, int32_t *order_out, float *UT_out, int32_t *KT_out);
/* Fetch tables/sorted columns for tests.  Returns 1 (not an error) when rows are uniform
 * and no tables exist. */
int gemhip_n2v_get_alias(gemhip_n2v_t h, float *U_host, int32_t *K_host, int32_t *col_sorted_host);
/* Walks start from the m nodes that occur in the edge list (the binary never sees an isolated node):
 * gemhip_n2v_start_nodes returns m.  SimulateWalk for global walk ids [walk_begin, walk_end) of m*num_walks
 * (walk id r*m + j starts at the j-th node of round r's permutation of those m nodes): this rank's shard. */
int gemhip_n2v_start_nodes(gemhip_n2v_t h, int64_t *m);
int gemhip_n2v_walks(gemhip_n2v_t h, float p, float q, int32_t num_walks, int32_t walk_len,
                     uint64_t seed, int32_t flags, int64_t walk_begin, int64_t walk_end,
                     void *stream);
int gemhip_n2v_set_walks(gemhip_n2v_t h, const int32_t *walks_host, int64_t nwalks,
                         int32_t walk_len, int64_t walk_id_offset);
int gemhip_n2v_get_walks(gemhip_n2v_t h, int32_t *walks_host);
int gemhip_n2v_walks_ptr(gemhip_n2v_t h, void **d_walks, int64_t *nwalks, int32_t *walk_len);
/* LearnVocab: token counts of the local walks into a device int32[n] (counts_ptr exposes
 * it so a multi-GPU driver can all-reduce it), then InitUnigramTable on the host. */
int gemhip_n2v_vocab(gemhip_n2v_t h, void *stream);
int gemhip_n2v_counts_ptr(gemhip_n2v_t h, void **d_counts);
/* Use a caller-owned DEVICE int32[n] as the count buffer (e.g. a torch tensor to all-reduce). */
int gemhip_n2v_bind_counts(gemhip_n2v_t h, void *d_counts);
int gemhip_n2v_build_unigram(gemhip_n2v_t h, int32_t *counts_out, float *UT_out, int32_t *KT_out);
/* InitPosEmb / InitNegEmb.  dSynPos/dSynNeg: optional caller-owned DEVICE tables
 * [n][d] float32 (both or neither). */
int gemhip_sgns_init(gemhip_n2v_t h, int32_t d, uint64_t seed, void *dSynPos, void *dSynNeg);
/* (centre, context) pairs trained since creation / the last reset -- the unit of SURVEY 8(d)'s
 * SGNS byte count (14*4d bytes per pair). */
int gemhip_sgns_pairs(gemhip_n2v_t h, int64_t *pairs, int32_t reset);
/* Cap on concurrently training wavefronts (Hogwild width).  0 = auto: the number of wavefronts for which the expected fraction of
 * row stores that overwrite another wavefront's store, rho = W x 5 x w / n_eff (w = the load-to-store window of a negative row in pair
 * steps: ~0.4 with reload-on-update, prefetch + 1 without; n_eff = 1 / sum q_v^2 over the negative-sampling distribution: n on a
 * uniform graph, far less with hubs), stays <= 1.5 % -- n_eff/133 with reload-on-update, n_eff/1000 without -- never more than the
 * device holds (on MI355X 1792 = seven per CU when every context row is cached, 1536 otherwise), over the cold rows only once hot rows take
 * atomic adds (gemhip_sgns_set_hot_rows), at most 2 % of the rows that occur; graphs below 8192 nodes: 1/16 of the table open at most.  Derivation and the
 * measurements behind it: DESIGN.md 3.3, scripts/hogwild_emul. */
int gemhip_n2v_set_max_waves(gemhip_n2v_t h, int32_t max_waves);
/* LDS window of TrainModel's context rows (no reference counterpart: the binary keeps its tables in host RAM).
 * radius = tokens either side of the centre word whose SynPos row stays in LDS between its first and last use
 * (-1 = auto = min(window, 10); 0 = off); delta_writeback: a row leaves the window as `row_now + (working - loaded)`
 * so that concurrent wavefronts' updates survive (1), or is written back as is (0); -1 = auto (1 unless one wavefront trains). */
int gemhip_sgns_set_window_cache(gemhip_n2v_t h, int32_t radius, int32_t delta_writeback);
/* How a Hogwild launch treats the rows it shares with other wavefronts (no reference counterpart: the binary's threads race on plain
 * doubles).  prefetch_pairs: (centre, context) pairs whose five negative rows are requested ahead of their use (2 default, 1; 0 = keep).
 * reload_on_update (1 default, 0, -1 = keep): apply a negative row's update to the row as it is at store time -- a second fetch right
 * after the dot products, `row_now + g * xc` -- and the centre's positive row as an atomic add of what the centre changed, instead of
 * storing copies that are (prefetch_pairs + 1) pair steps / one centre old and overwrite whatever other wavefronts stored meanwhile. */
int gemhip_sgns_set_hogwild(gemhip_n2v_t h, int32_t prefetch_pairs, int32_t reload_on_update);
/* Hot rows: a node with at least min_count tokens in the corpus never enters a wavefront's LDS window of context rows (its row is fetched
 * per pair and updated by atomic add) -- on a power-law graph a hub would otherwise sit in hundreds of windows at once and the deltas its
 * copies leave with add up.  -1 = auto: the nodes expected to be in another wavefront's window at any time, tokens / ((W-1) x (2R+1));
 * 0 = off.  No reference counterpart. */
int gemhip_sgns_set_hot_rows(gemhip_n2v_t h, int32_t min_count);
/* The launch gemhip_sgns_train would choose for one pass over `nwalks` walks of a corpus with these token counts (counts[n], as
 * gemhip_n2v_build_unigram returns them) on a handle with default knobs -- host arithmetic only, no device needed: which kernel (0 = no LDS
 * window, 1 = window with overwrite on leave, 2 = window with delta write-back: the Hogwild path), how many concurrent wavefronts (= walks
 * trained at once; the rule rho = W x 5 x w / n_eff <= 1.5 % of DESIGN.md 3.3), the hot-row token threshold (0: none), the effective table
 * size n_eff = 1 / sum q_v^2 of the unigram^0.75 distribution and the same over the cold rows only.  Any out pointer may be NULL.  No
 * reference counterpart (the binary runs as many threads as the machine has cores). */
int gemhip_sgns_plan_launch(const int32_t *counts, int64_t n, int32_t d, int32_t window, int32_t walk_len, int64_t nwalks, int32_t flags,
                            int32_t *kernel, int32_t *waves, int32_t *hot_threshold, double *n_eff, double *n_eff_cold);
/* What the LAST gemhip_sgns_train / gemhip_sgns_train_part on this handle actually launched -- the planner's choice after every knob and
 * environment override (GEMHIP_SGNS_MAX_WAVES, ...) and for the handle's own unigram-table layout: kernel (as gemhip_sgns_plan_launch), concurrent
 * wavefronts, hot-row token threshold, and the FRESH HOT ROWS bits in force (0 when the launch had no hot rows).  bench.py reports this instead of
 * replaying the planner.  Any out pointer may be NULL. */
int gemhip_sgns_last_launch(gemhip_n2v_t h, int32_t *kernel, int32_t *waves, int32_t *hot_threshold, int32_t *fresh);
/* LOCALLY HOT ROWS.  A node whose tokens are packed into few walks -- at least `per_walk` tokens per walk that contains it (default 8; 0 = off; -1 keeps the
 * handle's setting) -- sits in the LDS window of every such walk from its first token to its last, not for 2R+1 centres: the two nodes of an isolated edge
 * make up all 80 tokens of each of their 20 walks.  Two wavefronts training two of those walks at once each apply a whole walk's worth of updates to the
 * same base and both deltas are added (measured on R-MAT scale 17: one such pair ends with 14 % more norm than its peers and outranks the true neighbour of
 * dozens of them: 2-5 % of the graph's reconstruction MAP per event).  Hogwild launches of gemhip_sgns_train therefore treat these nodes as hot rows whatever
 * their token count (never cached; per-pair reads and atomic adds).  This call builds the key the kernels compare with the hot-row threshold from the walks
 * and counts on the handle -- hotkey[v] = INT32_MAX for a locally hot node, else its token count -- and returns how many nodes are locally hot and,
 * optionally, the key (n int32, host).  Needs the WHOLE corpus on the handle (a rank's shard against all-reduced counts would flag everything: the launch
 * skips the rule there).  No reference counterpart (the binary's threads read and write rows in place). */
int gemhip_n2v_locally_hot(gemhip_n2v_t h, int32_t per_walk, int64_t *count, int32_t *hotkey_host);
/* ... from a walk corpus assembled from every rank's shard (device pointer, corpus_rows x walk_len int32, rows of -1 = padding) and the handle's all-reduced
 * counts: the partitioned N-GPU schedule's form.  gemhip_sgns_train_part launches on the handle then treat the locally hot nodes as hot rows. */
int gemhip_n2v_locally_hot_corpus(gemhip_n2v_t h, const void *d_corpus, int64_t corpus_rows, int32_t walk_len, int32_t per_walk, int64_t *count, void *stream);
/* FRESH HOT ROWS (Hogwild launches that have hot rows; gem_amd/csrc/sgns.hpp SgnsArgs::fresh).  bit 0: a hot centre word's positive row takes every
 * pair's update as a returning atomic add and continues from the returned row; bit 1: hot negative rows are re-read right before the dot products.
 * bit 2 (value 4): every negative row with at least GEMHIP_SGNS_NEG_COUNT tokens is updated by atomic add instead of reload + store.
 * Bits 0 and 1 shorten the time between reading a hub row and adding a gradient computed from it (measured: the hubs' staleness falls 4x, the MAP gap does
 * not move: profiles/r06_*.jsonl); all three are A/B knobs, off by default.  No reference counterpart (the binary's Hogwild threads
 * read and write rows in place). */
int gemhip_sgns_set_fresh(gemhip_n2v_t h, int32_t bits);
/* Building block of the SGNS kernel, exposed for its own parity test: in[64][6] per-lane partial sums -> out[64], lane l
 * receiving the wave total of value (l & 4) ? 4 + (l & 1) : (l & 3). */
int gemhip_test_wave_sum6(const float *in_host, float *out_host);
int gemhip_sgns_set_tables(gemhip_n2v_t h, const float *SynPos_host, const float *SynNeg_host);
int gemhip_sgns_get_tables(gemhip_n2v_t h, float *SynPos_host, float *SynNeg_host);
/* TrainModel over LOCAL walks [walk_lo, walk_hi) for epoch `epoch` of `epochs`.
 * tokens_total = tokens of ALL ranks per epoch (the binary's AllWords), token_offset =
 * global word count before local walk 0 -- together they drive the linear alpha decay. */
int gemhip_sgns_train(gemhip_n2v_t h, int32_t window, int32_t neg, float alpha0,
                      int32_t epochs, int32_t epoch, int64_t walk_lo, int64_t walk_hi,
                      int64_t tokens_total, int64_t token_offset, uint64_t seed,
                      int32_t flags, void *stream);

/* Partitioned ("episode") SGNS for N GPUs -- no reference counterpart (the reference is one process); this is the
 * schedule that lets SGNS shard without two GPUs ever writing the same row: node v belongs to
 * partition v % parts with local row v / parts; rank g keeps SynPos partition g, the SynNeg partitions travel around a
 * ring, and in every round a rank trains ONE bucket (contexts of its SynPos partition, centre words of the visiting
 * SynNeg partition) of the walk corpus.  build_unigram_parts: one alias table per partition over local indices
 * (the unigram^0.75 distribution restricted to the partition), from the global counts of the handle. */
int gemhip_n2v_build_unigram_parts(gemhip_n2v_t h, int32_t parts, float *UT_out, int32_t *KT_out);
/* ... in the binary's layout (GEMHIP_N2V_VOCAB_ORDER): partition p = its nodes that occur, in order of first appearance in d_corpus (device pointer: the
 * walks of all ranks in walk-id order, -1 tokens = padding, corpus_tokens int32 entries; NULL = the handle's own walks).  Optional host outputs by local
 * row (UT_out / KT_out [n]), the slot tables (slot_out [n]: partition p's at its offset, -1 beyond its slot count) and the slot counts (nslots_out [parts]). */
int gemhip_n2v_build_unigram_parts_vocab_order(gemhip_n2v_t h, int32_t parts, int32_t flags, const void *d_corpus, int64_t corpus_tokens,
                                               float *UT_out, int32_t *KT_out, int32_t *slot_out, int64_t *nslots_out);
/* One bucket of the partitioned schedule trained in WALK order (round 4; no reference counterpart beyond TrainModel itself): TrainModel
 * (ELF @0x40d6a0) over a walk corpus in device memory, restricted to the pairs whose context is a node of partition ctx_part (rows of
 * dSynPos_part, local index v / parts) and whose centre word is a node of partition word_part (rows of dSynNeg_part; negatives from the
 * unigram table restricted to word_part -- gemhip_n2v_build_unigram_parts; the handle's counts must be the GLOBAL ones).
 * Corpus: nwalks work items.  d_seg == NULL: item i is row i of d_walks, walk id walk_id_offset + i.  Otherwise the corpus is assembled
 * from nseg shards (d_seg = device int64[3 * nseg] = {first row in d_walks, walks present, global id of the first walk} per shard) and item
 * i = r * seg_len + j is walk j of shard r (skipped when j >= walks present); nwalks must equal nseg * seg_len.
 * alpha = alpha0 * max(1 - t / (alpha_tokens_total + 1), 1e-4), t = token_offset + i * walk_len + position, refreshed every 10000 tokens
 * like the binary.  Same kernel and same draws per (walk id, position) as gemhip_sgns_train; over all parts x parts buckets every pair of
 * TrainModel is trained exactly once. */
int gemhip_sgns_train_part(gemhip_n2v_t h, const void *d_walks, int64_t nwalks, int32_t walk_len, const void *d_seg, int32_t nseg,
                           int64_t seg_len, int64_t walk_id_offset, int32_t window, float alpha0, int64_t alpha_tokens_total,
                           int64_t token_offset, int32_t epoch, uint64_t seed, int32_t flags, int32_t ctx_part, int32_t word_part,
                           void *dSynPos_part, void *dSynNeg_part, int32_t d, void *stream);
/* Local walks [walk_lo, walk_hi) copied (device to device, on `stream`) into a caller-owned device buffer. */
int gemhip_n2v_copy_walks(gemhip_n2v_t h, int64_t walk_lo, int64_t walk_hi, void *d_dst, void *stream);

/* ------------------------------------------------------------ N GPUs, one process (RCCL over xGMI)
 * the one-shot drop-ins above with an `n_gpus` argument.  The reference has no multi-device path; these shard the way
 * north_star prescribes -- GF by source row, node2vec by start node -- inside ONE host process that drives n_gpus devices, with the
 * collectives on RCCL (loaded with dlopen on first use: GEMHIP_E_UNSUPPORTED when librccl.so.1 is not there; the single-GPU entry points
 * never need it).  devices: n_gpus device ordinals, NULL = 0 .. n_gpus-1.  A list that names the SAME device n_gpus times runs the ranks as
 * VIRTUAL ranks on that device (test mode: RCCL refuses two ranks on one GPU, the collectives become device-to-device copies; sharding,
 * schedule and kernels are the production ones).  gem_amd/multi_gpu.py is the one-process-per-GPU form of the same schedules.
 *
 * gemhip_gf_train_multi: gemhip_gf_train with the source rows in n_gpus contiguous blocks, an in-place all-gather of the owned row blocks
 * after EVERY sweep -- bit-identical to one GPU (needs the firing sources in ascending id order when n_gpus > 1: GEMHIP_E_UNSUPPORTED
 * otherwise).  stats (optional, 8 doubles): {sweep + exchange seconds, updates per sweep, rows per sweep, exchange bytes per rank per
 * sweep, n_gpus, virtual (0/1), 0, 0}.
 * gemhip_n2v_train_multi: gemhip_n2v_train with walks by start-node shard, an all-reduce of the token counts, one all-gather of the walk
 * shards and the partitioned-table schedule (`episodes` slices of every shard; rank g trains bucket (g, (g+s) % n_gpus) of each with
 * gemhip_sgns_train_part and passes its SynNeg partition around a ring).  stats (optional, 8 doubles): {walk + vocabulary + gather
 * seconds, training seconds, tokens, pairs trained, ring bytes per rank per round, n_gpus, virtual (0/1), bucket launches per rank}.
 * Unigram-table layout: flags with GEMHIP_N2V_VOCAB_ORDER (bit 16, the plugin default) lay every partition's alias table out over its nodes in order
 * of first appearance in the whole corpus (gemhip_n2v_build_unigram_parts_vocab_order: the binary's layout restricted to the partition; with one
 * device it IS gemhip_n2v_train's table, and the deterministic mode gives the same embedding bit for bit); without the bit, in node-id order
 * (gemhip_n2v_build_unigram_parts).  Round 4 ignored the bit here.
 * STATUS of n_gpus > 1 on DISTINCT devices: the RCCL path (grouped in-place ncclAllGather, ncclAllReduce, ncclSend/ncclRecv ring, one
 * non-blocking stream per device, all driven from one host thread) has only ever run with ONE rank (a 1-GPU test pool); with n_gpus > 1
 * it has been exercised as virtual ranks only, where the three collectives are copies on one stream.  Treat the multi-device path as
 * unvalidated until tests/test_multi_capi_gpu.py::test_real_devices_* has run on a box with >= 2 GPUs (they skip otherwise).
 * gemhip_rccl_selftest: communicator create -> all-gather / all-reduce / ring shift of `bytes` per rank on known patterns, every word
 * checked -> destroy. */
int gemhip_gf_train_multi(int64_t n, int64_t m, const int32_t *src, const int32_t *dst, const float *w, int32_t d, float eta,
                          float regu, int32_t max_iter, int32_t n_gpus, const int32_t *devices, float *X_inout, double *stats);
int gemhip_n2v_train_multi(int64_t n, int64_t nnz, const int64_t *row_ptr, const int32_t *col, const float *w, int32_t d,
                           int32_t walk_len, int32_t num_walks, int32_t window, int32_t epochs, float p, float q, uint64_t seed,
                           int32_t flags, int32_t n_gpus, const int32_t *devices, int32_t episodes, float *X_out, double *stats);
int gemhip_rccl_selftest(int32_t n_gpus, const int32_t *devices, int64_t bytes, double *seconds);

/* ---------------------------------------------------------------------- HOPE
 * Replaces: gem/embedding/hope.py:23-41 (HOPE.learn_embedding): S = inv(I - beta A) (beta A)
 * as dense numpy matrices (:28-31) and u, s, vt = scipy.sparse.linalg.svds(S, k=d//2) (:33).
 *
 * Graph: CSR of A with rows AND columns indexed in graph.nodes order (hope.py:28 uses
 * nx.to_numpy_matrix, whose index is insertion order).  Outputs, all caller-owned float32:
 *   U_sqrtS [n][k] = u * sqrt(s),  V_sqrtS [n][k] = vt.T * sqrt(s)   (hope.py:34-35; X = [U | V]),
 *   sigma [k] ASCENDING like svds (hope.py:33).  Column signs: largest |entry| of each u positive.
 * Solver knobs: oversample (block = k + oversample columns), krylov_steps (Krylov blocks per cycle until pairs
 * start to lock; converged leading pairs are frozen and the freed basis columns -- capacity 1.2 * (krylov_steps + 1)
 * blocks -- deepen the polynomial for the rest), max_restarts, tol (max relative change of the k singular values
 * between cycles; pairs lock at a relative residual of 0.1 * sqrt(tol)).
 * stats (optional, 12 doubles): {device_seconds, spmm_launches, spmm_columns_total, katz_terms,
 * basis_columns, restarts_done, last_sigma_change, beta*sigma_max(A) estimate, host_eig_seconds,
 * host_eig_calls, max relative Ritz residual of the previous cycle,
 * seconds inside SpMM launches (HIP events)}.
 * Symmetric A (A == A^T entry for entry, rows sorted by column, beta > 0, n >= 16384; GEMHIP_HOPE_SYM=0/1 forces it off/on):
 * S = f(A), f(x) = beta x / (1 - beta x), shares A's eigenvectors, so the same triplets -- sigma = |f(lambda)|, v = q,
 * u = sign(f(lambda)) q -- come from a Chebyshev-filtered subspace iteration on A itself (one SpMM per polynomial degree,
 * block-sized Rayleigh-Ritz, locked pairs deflated inside the filter); krylov_steps is unused there, max_restarts bounds the
 * cycles at max(40, 3 * max_restarts), and a run that does not converge is redone by the block-Krylov solver.  stats then
 * carries katz_terms = 0 and restarts_done = filter cycles.
 * Returns GEMHIP_E_NOTCONVERGED when beta*sigma_max(A) >= 0.95 (Katz series too slow). */
/* Staged form: the graph-dependent setup (A and A^T in CSR on the device, Katz series length) is done once. */
typedef struct gemhip_hope_plan *gemhip_hope_plan_t;
int gemhip_hope_plan_create(int64_t n, int64_t nnz, const int64_t *row_ptr, const int32_t *col, const float *w,
                            float beta, gemhip_hope_plan_t *out);
int gemhip_hope_plan_solve(gemhip_hope_plan_t plan, int32_t k, int32_t oversample, int32_t krylov_steps,
                           int32_t max_restarts, float tol, uint64_t seed, float *U_sqrtS, float *V_sqrtS,
                           float *sigma, double *stats);
/* gemhip_hope_plan_solve with U sqrt(S) / V sqrt(S) written to DEVICE memory (n x k floats each; sigma, stats: host): the embedding stays in HBM
 * for a caller that evaluates or post-processes it there.  No reference counterpart (hope.py returns numpy arrays). */
int gemhip_hope_plan_solve_device(gemhip_hope_plan_t P, int32_t k, int32_t oversample, int32_t krylov_steps, int32_t max_restarts, float tol,
                                  uint64_t seed, void *dU_sqrtS, void *dV_sqrtS, float *sigma, double *stats);
int gemhip_hope_plan_destroy(gemhip_hope_plan_t plan);
int gemhip_hope(int64_t n, int64_t nnz, const int64_t *row_ptr, const int32_t *col, const float *w,
                float beta, int32_t k, int32_t oversample, int32_t krylov_steps,
                int32_t max_restarts, float tol, uint64_t seed, float *U_sqrtS, float *V_sqrtS,
                float *sigma, double *stats);
/* hope.py:38-40 `print('SVD error (low rank): %f' % norm(u diag(s) vt - S))`: for the truncated SVD that matrix is S (I - V V^T), and its
 * squared Frobenius norm is estimated with `probes` (1..128; 32 is plenty) Gaussian probe columns, deflated by V, pushed through the Katz series by
 * the solver's own SpMM kernel, with the exactly known ||beta A (I - V V^T)||_F^2 as a control variate.  sigma / V_sqrtS: the k singular values and
 * the n x k block V sqrt(Sigma) a solve returned (host).  frob2_out (optional): err^2 + sum sigma^2, the estimate of ||S||_F^2. */
int gemhip_hope_plan_svd_error(gemhip_hope_plan_t plan, int32_t k, const float *sigma, const float *V_sqrtS, int32_t probes, uint64_t seed,
                               double *err_out, double *frob2_out);
int gemhip_hope_svd_error(int64_t n, int64_t nnz, const int64_t *row_ptr, const int32_t *col, const float *w, float beta, int32_t k,
                          const float *sigma, const float *V_sqrtS, int32_t probes, uint64_t seed, double *err_out, double *frob2_out);
/* ... with the U side: ||u diag(s) vt - S||_F^2 = ||S (I - V V^T)||_F^2 + ||S V - U Sigma||_F^2 for orthonormal V (the two parts are orthogonal); the second
 * term is computed exactly (the Katz series on V's k columns) and is ~0 for a converged solve, so a wrong or unconverged U shows in the number exactly as
 * it does in hope.py:38-40's.  The V-only forms above report the truncation error an exact SVD with this V would have.  Columns with sigma[j] <= 0 (k
 * above the rank of S) are skipped.  uside_out (optional): sqrt of the second term alone.  What HOPE(..., verbose=True) prints. */
int gemhip_hope_plan_svd_error_uv(gemhip_hope_plan_t plan, int32_t k, const float *sigma, const float *U_sqrtS, const float *V_sqrtS, int32_t probes,
                                  uint64_t seed, double *err_out, double *frob2_out, double *uside_out);
int gemhip_hope_svd_error_uv(int64_t n, int64_t nnz, const int64_t *row_ptr, const int32_t *col, const float *w, float beta, int32_t k,
                             const float *sigma, const float *U_sqrtS, const float *V_sqrtS, int32_t probes, uint64_t seed, double *err_out,
                             double *frob2_out, double *uside_out);

/* ------------------------------------------------ Laplacian Eigenmaps ("next")
 * Replaces gem/embedding/lap.py:21-37: eigs(nx.normalized_laplacian_matrix(graph.to_undirected()), k=d+1, which='SM').
 * Input: CSR of the SYMMETRIC weighted adjacency in graph.nodes order (w NULL = unit).  The k smallest eigenpairs of
 * L_sym = I - D^-1/2 A D^-1/2 are computed as the k largest of I + D^-1/2 A D^-1/2 with the HOPE block-Krylov solver
 * (one SpMM per operator application).  V_out [n][k]: unit eigenvectors, eigvals[k] ascending (column 0 is the
 * trivial eigenvector lap.py drops with v[:, 1:]).  stats: as gemhip_hope.  From 16384 nodes up (GEMHIP_HOPE_SYM=0/1 forces it
 * off/on) the Chebyshev-filtered eigen-path of gemhip_hope is used (stats katz_terms = -1), with the block-Krylov solver as
 * the fallback when it does not converge. */
int gemhip_lap_eigmap(int64_t n, int64_t nnz, const int64_t *row_ptr, const int32_t *col, const float *w,
                      int32_t k, int32_t oversample, int32_t krylov_steps, int32_t max_restarts, float tol,
                      uint64_t seed, float *V_out, float *eigvals, double *stats);

/* Locally Linear Embedding.  Replaces gem/embedding/lle.py:23-35: svds(I - D^-1 A, k=d+1, which='SM').
 * Input as gemhip_lap_eigmap (symmetric adjacency; rows are l1-normalised here).  sing[k]: the k smallest singular
 * values of I - P ascending; V_out [n][k]: matching right singular vectors (column 0 ~ the constant vector lle.py drops).
 * Same size switch as gemhip_lap_eigmap: the eigen-path filters N^T N, N = I - P, towards its smallest eigenvalues
 * (stats katz_terms = -2). */
int gemhip_lle(int64_t n, int64_t nnz, const int64_t *row_ptr, const int32_t *col, const float *w, int32_t k,
               int32_t oversample, int32_t krylov_steps, int32_t max_restarts, float tol, uint64_t seed,
               float *V_out, float *sing, double *stats);

/* HOPE building blocks, exposed so each kernel can be parity-tested on its own (host buffers in,
 * host buffers out, blocking).  Dense blocks are row-major with leading dimension = column count.
 *   sym_eig : host fp64 symmetric eigensolver used for the projected problems (A overwritten by
 *             eigenvectors in columns, w ascending) -- pure host code, callable without a GPU;
 *   spmm    : Y[n][b] = alpha * A X (+ Wadd)        gram : G[m1][m2] = X^T Y  (fp64 out, MFMA fp32)
 *   tsgemm  : Out[n][b2] = (Src or 0) + alpha * X[n][m] C[m][b2]   (C given in fp64, MFMA fp32). */
int gemhip_sym_eig(int32_t n, double *A_inout, double *w_out);
int gemhip_sym_eig_builtin(int32_t n, double *A_inout, double *w_out);
/* The m largest eigenpairs only (what the Rayleigh-Ritz step consumes): Householder reduction, QL eigenvalues, inverse
 * iteration, back-transformation of m vectors.  A destroyed; w_out: m eigenvalues DESCENDING; Z_out: m eigenvectors,
 * one after the other (n doubles each).  Host code, callable without a GPU. */
int gemhip_sym_eig_top(int32_t n, double *A_inout, int32_t m, double *w_out, double *Z_out);
/* Host threads of the built-in eigensolver's O(n^3) phases (Householder reduction from n = 192 on, back-transformation of the
 * wanted vectors).  threads >= 1 (at most 16) sets the count, <= 0 restores the default: GEMHIP_EIG_THREADS, else
 * min(4, half the cores this process may run on).  in_effect_out (may be NULL) receives the count in effect.  Results of the
 * reduction agree between thread counts to rounding (sums are taken in a different order), not bit for bit.  There is no
 * reference counterpart: hope.py:32 hands the whole SVD to ARPACK/LAPACK, which thread through their BLAS. */
int gemhip_set_host_threads(int32_t threads, int32_t *in_effect_out);
/* Optional: let the host supply a faster symmetric eigensolver for the projected problems (same contract as
 * gemhip_sym_eig: row-major symmetric A overwritten by eigenvectors in columns, w ascending, return 0).  The Python
 * layer registers numpy's LAPACK (dsyevd) here; NULL restores the built-in Householder/QL solver. */
int gemhip_set_sym_eig_callback(int (*fn)(int32_t n, double *A_inout, double *w_out));
int gemhip_hope_spmm(int64_t n, int64_t nnz, const int64_t *row_ptr, const int32_t *col, const float *w,
                     float alpha, int32_t b, const float *X_host, const float *Wadd_host, float *Y_host);
int gemhip_hope_gram(int64_t n, int32_t m1, int32_t m2, const float *X_host, const float *Y_host,
                     double *G_host);
int gemhip_hope_tsgemm(int64_t n, int32_t m, int32_t b2, const float *X_host, const double *C_host,
                       float alpha, const float *Src_host, float *Out_host);

/* -------------------------------------------------- evaluation
 * Average precision of graph reconstruction for SAMPLED nodes, with the exact semantics of
 * gem/evaluation/metrics.py:27-46 (computeMAP) over the candidate list of
 * gem/utils/evaluation_util.py:28-35 (j > i when undirected, weight > 0), where the weight is
 * score(i, j) = A_i . B_j  (A = B = X for GF / node2vec, gf.py:103-104; A = X[:, :k], B = X[:, k:] for
 * HOPE, hope.py:43-44) -- without the n x n matrix of static_graph_embedding.py:48-65.
 * A_host, B_host: [n][ld] float32 (B_host NULL = A); row_ptr/col: CSR of the TRUE graph; ap_out[nsample].
 * MAP over the sample = mean(ap_out).  Any degree: a hub's true neighbours are processed in chunks of 512
 * (one workgroup per chunk); duplicate columns in the CSR count once. */
int gemhip_eval_sampled_ap(int64_t n, int32_t da, int32_t ld, const float *A_host, const float *B_host,
                           const int64_t *row_ptr, const int32_t *col, int32_t undirected,
                           int64_t nsample, const int32_t *nodes, double *ap_out);

#ifdef __cplusplus
}
#endif
#endif /* GEM_HIP_H */
