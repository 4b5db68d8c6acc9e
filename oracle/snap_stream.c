/*
 * oracle/snap_stream.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * C port of oracle/snap_stream.py (read that file's header first): the reference's node2vec path -- the prebuilt SNAP
 * binary gem/c_exe/node2vec run by gem/embedding/node2vec.py:34-48 -- restated ON ITS OWN RANDOM STREAM (TRnd) and in
 * its own fp64, so that the restatement can be compared with the binary itself (made deterministic by
 * oracle/shim/faketime.c + OMP_NUM_THREADS=1) at sizes the pure-Python loops cannot reach: the reference's default
 * hyper-parameters d = 128, r = 10, l = 80, k = 10 on SBM-1024 (tests/golden/n2v_snap_stream_sbm1024.npz).
 * tests/test_oracle_n2v.py checks (i) this port against snap_stream.py on the small cases, token for token and number
 * for number, and (ii) this port against the binary's walk matrix (hash) and embedding file at that size.
 * ELF addresses of what is restated: see snap_stream.py.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef struct { int32_t seed; } trnd;
static int32_t trnd_next(trnd *r)
{
    const int32_t s = r->seed;
    int32_t v = 16807 * (s % 127773) - 2836 * (s / 127773);
    if (v <= 0) v += 2147483647;
    r->seed = v;
    return v;
}
static double trnd_uni(trnd *r) { return (double)trnd_next(r) / 2147483647.0; }
static int32_t trnd_int(trnd *r, int32_t range) { const int32_t v = trnd_next(r); return range == 0 ? v : v % range; }

/* GetNodeAlias / InitUnigramTable's second half: P normalised (fp64) -> K, U; stacks filled in index order, popped from the back */
static void vose(int64_t N, const double *P, int32_t *K, double *U, int32_t *under, int32_t *over)
{
    int64_t nu = 0, no = 0;
    for (int64_t i = 0; i < N; ++i) {
        K[i] = 0;
        U[i] = P[i] * (double)N;
        if (U[i] < 1) under[nu++] = (int32_t)i; else over[no++] = (int32_t)i;
    }
    while (nu > 0 && no > 0) {
        const int32_t s = under[--nu], l = over[--no];
        K[s] = l;
        U[l] = U[l] + U[s] - 1;
        if (U[l] < 1) under[nu++] = l; else over[no++] = l;
    }
    while (nu > 0) U[under[--nu]] = 1.0;
    while (no > 0) U[over[--no]] = 1.0;
}

static int has_edge(const int64_t *row_ptr, const int32_t *col, int32_t t, int32_t x)
{
    int64_t lo = row_ptr[t], hi = row_ptr[t + 1];
    while (lo < hi) { const int64_t mid = (lo + hi) >> 1; if (col[mid] < x) lo = mid + 1; else hi = mid; }
    return lo < row_ptr[t + 1] && col[lo] == x;
}

/* node2vec(): PreprocessTransitionProbs, then num_walks rounds of Shuffle + SimulateWalk.
 * order[m]: node ids in order of first appearance in the edge file; CSR over ids [0, n) with SORTED columns, w per CSR entry.
 * out[num_walks * m][walk_len], zero padded.  Returns 0, or -1 when out of memory. */
int snap_stream_walks(int64_t n, int64_t m, const int32_t *order, const int64_t *row_ptr, const int32_t *col, const double *w,
                      double p, double q, int32_t num_walks, int32_t walk_len, int32_t seed, int32_t *out)
{
    const int64_t nnz = row_ptr[n];
    /* table of the pair (t -> v) = CSR entry e: deg(v) slots at toff[e] */
    int64_t *toff = (int64_t *)malloc(sizeof(int64_t) * (size_t)(nnz + 1));
    if (!toff) return -1;
    int64_t total = 0, maxdeg = 1;
    for (int64_t t = 0; t < n; ++t)
        for (int64_t e = row_ptr[t]; e < row_ptr[t + 1]; ++e) {
            const int32_t v = col[e];
            const int64_t dv = row_ptr[v + 1] - row_ptr[v];
            toff[e] = total; total += dv;
            if (dv > maxdeg) maxdeg = dv;
        }
    toff[nnz] = total;
    int32_t *K = (int32_t *)malloc(sizeof(int32_t) * (size_t)(total + 1));
    double *U = (double *)malloc(sizeof(double) * (size_t)(total + 1));
    double *P = (double *)malloc(sizeof(double) * (size_t)maxdeg);
    int32_t *under = (int32_t *)malloc(sizeof(int32_t) * (size_t)maxdeg), *over = (int32_t *)malloc(sizeof(int32_t) * (size_t)maxdeg);
    int32_t *ids = (int32_t *)malloc(sizeof(int32_t) * (size_t)(m > 0 ? m : 1));
    if (!K || !U || !P || !under || !over || !ids) { free(toff); free(K); free(U); free(P); free(under); free(over); free(ids); return -1; }
    for (int64_t t = 0; t < n; ++t)
        for (int64_t e = row_ptr[t]; e < row_ptr[t + 1]; ++e) {
            const int32_t v = col[e];
            const int64_t a = row_ptr[v], dv = row_ptr[v + 1] - a;
            double psum = 0;
            for (int64_t j = 0; j < dv; ++j) {
                const int32_t x = col[a + j];
                const double wt = w[a + j];
                const double pr = (x == (int32_t)t) ? wt / p : (has_edge(row_ptr, col, (int32_t)t, x) ? wt : wt / q);
                P[j] = pr; psum += pr;
            }
            for (int64_t j = 0; j < dv; ++j) P[j] /= psum;
            vose(dv, P, K + toff[e], U + toff[e], under, over);
        }
    memset(out, 0, sizeof(int32_t) * (size_t)num_walks * (size_t)m * (size_t)walk_len);
    trnd rnd = {seed};
    for (int64_t i = 0; i < m; ++i) ids[i] = order[i];
    for (int32_t r = 0; r < num_walks; ++r) {
        for (int64_t i = 0; i + 1 < m; ++i) {                     /* TVec::Shuffle */
            const int64_t j = i + trnd_int(&rnd, (int32_t)(m - i));
            const int32_t tmp = ids[i]; ids[i] = ids[j]; ids[j] = tmp;
        }
        for (int64_t j = 0; j < m; ++j) {
            int32_t *wk = out + ((int64_t)r * m + j) * walk_len;
            int32_t len = 0;
            const int32_t s = ids[j];
            wk[len++] = s;
            const int64_t ds = row_ptr[s + 1] - row_ptr[s];
            if (walk_len == 1 || ds == 0) continue;
            int64_t e_prev = row_ptr[s] + trnd_int(&rnd, (int32_t)ds);     /* CSR entry of the edge (src -> dst) just taken */
            wk[len++] = col[e_prev];
            while (len < walk_len) {
                const int32_t dst = wk[len - 1];
                const int64_t a = row_ptr[dst], dd = row_ptr[dst + 1] - a;
                if (dd == 0) break;
                const int32_t *Kt = K + toff[e_prev];
                const double *Ut = U + toff[e_prev];
                const int64_t x = (int64_t)(trnd_uni(&rnd) * (double)dd);
                const double y = trnd_uni(&rnd);
                const int64_t nx = (y < Ut[x]) ? x : Kt[x];
                e_prev = a + nx;
                wk[len++] = col[e_prev];
            }
        }
    }
    free(toff); free(K); free(U); free(P); free(under); free(over); free(ids);
    return 0;
}

/* LearnEmbeddings on one thread.  walks[nwalks][walk_len]: raw ids in [0, id_bound) on entry, RENAMED ids on return.
 * ids_out[id_bound], emb_out[id_bound][d]: the first *n_out rows are the output file's rows, in its order.  Returns 0 / -1. */
int snap_stream_learn_embeddings(int64_t nwalks, int32_t walk_len, int32_t *walks, int64_t id_bound, int32_t d, int32_t window,
                                 int32_t iters, int32_t seed, int32_t rename, int32_t *ids_out, double *emb_out, int64_t *n_out)
{
    /* rename = 1: the binary.  rename = 0 (an experiment, scripts/unigram_layout_effect.py): tokens keep their node ids, i.e. the
     * vocabulary and with it the unigram alias table are laid out in node-id order -- the layout of oracle/n2v_oracle.c and of the HIP
     * path.  Under RndUnigramInt's quirk the negative distribution depends on that layout. */
    enum { MAX_EXP = 6, PRECISION = 10000, TABLE = MAX_EXP * PRECISION * 2, NEG = 5 };
    const double start_alpha = 0.025;
    int32_t *rnm = (int32_t *)malloc(sizeof(int32_t) * (size_t)id_bound);
    if (!rnm) return -1;
    for (int64_t i = 0; i < id_bound; ++i) rnm[i] = -1;
    int64_t N = 0;
    if (!rename) { for (int64_t i = 0; i < id_bound; ++i) { rnm[i] = (int32_t)i; ids_out[i] = (int32_t)i; } N = id_bound; }
    const int64_t all_words = nwalks * walk_len;
    for (int64_t i = 0; i < all_words; ++i) {
        const int32_t v = walks[i];
        if (rnm[v] < 0) { rnm[v] = (int32_t)N; ids_out[N++] = v; }
        walks[i] = rnm[v];
    }
    int64_t *vocab = (int64_t *)calloc((size_t)N, sizeof(int64_t));
    double *syn_pos = emb_out, *syn_neg = (double *)calloc((size_t)N * (size_t)d, sizeof(double));
    double *P = (double *)malloc(sizeof(double) * (size_t)N), *U = (double *)malloc(sizeof(double) * (size_t)N);
    int32_t *K = (int32_t *)malloc(sizeof(int32_t) * (size_t)N), *under = (int32_t *)malloc(sizeof(int32_t) * (size_t)N),
            *over = (int32_t *)malloc(sizeof(int32_t) * (size_t)N);
    double *exp_table = (double *)malloc(sizeof(double) * TABLE), *neu1e = (double *)malloc(sizeof(double) * (size_t)d);
    if (!vocab || !syn_neg || !P || !U || !K || !under || !over || !exp_table || !neu1e) return -1;
    for (int64_t i = 0; i < all_words; ++i) ++vocab[walks[i]];
    trnd rnd = {seed};
    for (int64_t i = 0; i < N; ++i)
        for (int32_t j = 0; j < d; ++j) syn_pos[i * d + j] = (trnd_uni(&rnd) - 0.5) / (double)d;
    double tot = 0;
    for (int64_t i = 0; i < N; ++i) { P[i] = pow((double)vocab[i], 0.75); tot += P[i]; }
    for (int64_t i = 0; i < N; ++i) P[i] /= tot;
    vose(N, P, K, U, under, over);
    for (int32_t i = 0; i < TABLE; ++i) exp_table[i] = pow(2.718281828459045235360287, -MAX_EXP + (double)i / (double)PRECISION);
    double alpha = start_alpha;
    int64_t cnt = 0;
    for (int32_t it = 0; it < iters; ++it)
        for (int64_t wi = 0; wi < nwalks; ++wi) {
            const int32_t *wk = walks + wi * walk_len;
            for (int32_t pos = 0; pos < walk_len; ++pos) {
                if (cnt % 10000 == 0) {
                    alpha = start_alpha * (1 - (double)cnt / (double)((int64_t)iters * all_words + 1));
                    if (alpha < start_alpha * 0.0001) alpha = start_alpha * 0.0001;
                }
                const int32_t word = wk[pos];
                const int32_t off = trnd_int(&rnd, 0) % window;
                for (int32_t a = off; a < window * 2 + 1 - off; ++a) {
                    if (a == window) continue;
                    const int32_t cp = pos - window + a;
                    if (cp < 0 || cp >= walk_len) continue;
                    double *xc = syn_pos + (int64_t)wk[cp] * d;
                    for (int32_t k = 0; k < d; ++k) neu1e[k] = 0;
                    for (int32_t j = 0; j < NEG + 1; ++j) {
                        int32_t target, label;
                        if (j == 0) { target = word; label = 1; }
                        else {
                            const int32_t X = K[(int64_t)(trnd_uni(&rnd) * (double)N)];
                            const double Y = trnd_uni(&rnd);
                            target = (Y < U[X]) ? X : K[X];
                            if (target == word) continue;
                            label = 0;
                        }
                        double *yt = syn_neg + (int64_t)target * d;
                        double prod = 0;
                        for (int32_t k = 0; k < d; ++k) prod += xc[k] * yt[k];
                        double g;
                        if (prod > MAX_EXP) g = (label - 1) * alpha;
                        else if (prod < -MAX_EXP) g = label * alpha;
                        else g = (label - 1 + 1 / (1 + exp_table[(int)(prod * PRECISION) + TABLE / 2])) * alpha;
                        for (int32_t k = 0; k < d; ++k) { neu1e[k] += g * yt[k]; yt[k] += g * xc[k]; }
                    }
                    for (int32_t k = 0; k < d; ++k) xc[k] += neu1e[k];
                }
                ++cnt;
            }
        }
    *n_out = N;
    free(rnm); free(vocab); free(syn_neg); free(P); free(U); free(K); free(under); free(over); free(exp_table); free(neu1e);
    return 0;
}
