/*
 * scripts/hogwild_emul/hogwild_emul.c -- INVESTIGATION TOOL (round 3), neither product nor oracle.
 *
 * Question ("what's weak" #3): sgns_win_kernel loses ~0.07 % of reconstruction MAP per 100 concurrent wavefronts
 * at SBM 1M/10M.  WHICH of the kernel's departures from the sequential TrainModel costs it?  This file replays TrainModel
 * (the arithmetic of oracle/n2v_oracle.c::oracle_sgns_train, included below for its Philox / helpers) with W VIRTUAL
 * WAVEFRONTS that advance round-robin, one (centre, context) pair per turn, each with the private row copies the kernel holds:
 *
 *   centre row   SynNeg[word]   : loaded at the centre's first pair, stored after its last          (kernel: registers `yp`)
 *   context rows SynPos[ctx]    : enter a (2R+1)-token window cache, leave it R centres later       (kernel: LDS window)
 *   negative rows SynNeg[tgt]   : loaded L pairs before they are used, stored right after use        (kernel: q0/q1/q2 prefetch)
 *
 * and a MODE per class:  0 = direct (read-modify-write on the shared table inside the turn: no staleness, nothing lost)
 *                        1 = private copy, OVERWRITE on store (stale gradient + foreign updates in between are LOST)
 *                        2 = private copy, DELTA on store: shared += (working - loaded) (stale gradient, nothing lost)
 * With W = 1 every mode is exactly oracle_sgns_train (tests: scripts/hogwild_emul/run.py --selftest).
 * Walk -> wavefront assignment is the kernel's: wavefront g trains walks g, g+W, g+2W, ...
 */
#include "../../oracle/n2v_oracle.c"

#define MAXQ 64      /* pairs generated ahead per wavefront (one centre has <= 2*window) */
#define MAXL 8
#define NEG 5

typedef struct {
    int64_t wl; int32_t word, ctx, pos; int32_t tgt[NEG]; float alpha; int first, last;
    float *ny;       /* [NEG][d] private copies (as used)  */
    float *nl;       /* [NEG][d] as loaded                 */
    int loaded;
} Pair;

typedef struct { int32_t node; int ref; float *work, *orig; } Slot;

typedef struct {
    int64_t wl; int pos; int walk_live;
    Pair q[MAXQ]; int qh, qn;          /* ring of generated pairs: q[qh .. qh+qn) */
    float *yp, *ypl; int32_t cword;    /* centre row copy */
    Slot *slots; int nslots; int win_pos;   /* window cache state: centre position it is arranged for */
    float *pool;
} Wave;

typedef struct {
    int64_t n; int32_t d; int64_t nwalks; int32_t walk_len; const int32_t *walks; int32_t window; float alpha0;
    int64_t denom; int64_t token_offset; int64_t walk_id_offset; int32_t epoch; const float *UT; const int32_t *KT; uint64_t seed; int32_t flags;
    float *SynPos, *SynNeg; int W, L, R, ctr_mode, ctx_mode, neg_mode;
    const int32_t *counts; int32_t hot_thr;      /* nodes with counts[v] >= hot_thr > 0: never cached as contexts, negative updates applied at store time */
    int64_t stat_pairs, stat_neg_lost, stat_ctr_lost;
} Cfg;

static const int32_t *g_slot_tab = NULL; static int64_t g_nslots = 0;
void hogwild_emul_set_slot_table(const int32_t *slot_tab, int64_t n_slots) { g_slot_tab = slot_tab; g_nslots = n_slots; }

/* generate the pairs of the next non-empty centre of wave w into its ring; returns 0 when the wave has no walks left */
static int gen_centre(const Cfg *c, Wave *w)
{
    while (1) {
        if (!w->walk_live) return 0;
        if (w->pos >= c->walk_len) {
            w->wl += c->W; w->pos = 0;
            if (w->wl >= c->nwalks) { w->walk_live = 0; return 0; }
        }
        const int32_t *walk = c->walks + w->wl * c->walk_len;
        const int pos = w->pos++;
        const int32_t word = walk[pos];
        if (word < 0) continue;
        const int64_t wid = c->walk_id_offset + w->wl;
        const int64_t t = c->token_offset + w->wl * c->walk_len + pos;
        const float alpha = sgns_alpha(c->alpha0, t, c->denom);
        const u32x4 rw = philox(c->seed, (uint32_t)wid, (uint32_t)((uint64_t)wid >> 32), (uint32_t)pos, TAG_WIN | ((uint32_t)c->epoch << 8));
        const int32_t b = (int32_t)(rw.x % (uint32_t)c->window);
        int made = 0;
        for (int32_t a = b; a < c->window * 2 + 1 - b; ++a) {
            if (a == c->window) continue;
            const int32_t cp = pos - c->window + a;
            if (cp < 0 || cp >= c->walk_len) continue;
            const int32_t ctx = walk[cp];
            if (ctx < 0) continue;
            Pair *p = &w->q[(w->qh + w->qn) % MAXQ];
            p->wl = w->wl; p->word = word; p->ctx = ctx; p->pos = pos; p->alpha = alpha; p->first = (made == 0); p->last = 0; p->loaded = 0;
            for (int j = 1; j <= NEG; ++j) {
                const u32x4 rn = philox(c->seed, (uint32_t)wid, (uint32_t)((uint64_t)wid >> 32), (uint32_t)pos | ((uint32_t)a << 16),
                                        TAG_NEG | ((uint32_t)c->epoch << 8) | ((uint32_t)j << 16));
                /* the binary's table layout (round 5; hogwild_emul_set_slot_table): slots over the nodes that occur, UT / KT indexed by node -- as in sgns_train_core */
                const uint32_t slot = mulhi_range(rn.x, (uint32_t)(g_slot_tab ? g_nslots : c->n));
                const int32_t X = g_slot_tab ? g_slot_tab[slot] : (c->flags & 2) ? c->KT[slot] : (int32_t)slot;
                p->tgt[j - 1] = (u01(rn.y) < c->UT[X]) ? X : c->KT[X];
            }
            ++w->qn; ++made;
        }
        if (made) { w->q[(w->qh + w->qn - 1) % MAXQ].last = 1; return 1; }
    }
}

static void slot_writeback(Cfg *c, Slot *s)
{
    float *g = c->SynPos + (size_t)s->node * c->d;
    if (c->ctx_mode == 1) memcpy(g, s->work, sizeof(float) * c->d);
    else for (int k = 0; k < c->d; ++k) g[k] = g[k] + (s->work[k] - s->orig[k]);
    s->node = -1; s->ref = 0;
}
static Slot *slot_find(Wave *w, int32_t node)
{
    for (int i = 0; i < w->nslots; ++i) if (w->slots[i].node == node) return &w->slots[i];
    return NULL;
}
static int is_hot(const Cfg *c, int32_t node) { return c->hot_thr > 0 && node >= 0 && c->counts[node] >= c->hot_thr; }
static void slot_enter(Cfg *c, Wave *w, int32_t node)
{
    if (node < 0 || is_hot(c, node)) return;
    Slot *s = slot_find(w, node);
    if (s) { ++s->ref; return; }
    for (int i = 0; i < w->nslots; ++i) if (w->slots[i].node < 0) { s = &w->slots[i]; break; }
    s->node = node; s->ref = 1;
    memcpy(s->work, c->SynPos + (size_t)node * c->d, sizeof(float) * c->d);
    memcpy(s->orig, s->work, sizeof(float) * c->d);
}
static void slot_leave(Cfg *c, Wave *w, int32_t node)
{
    if (node < 0 || is_hot(c, node)) return;
    Slot *s = slot_find(w, node);
    if (--s->ref == 0) slot_writeback(c, s);
}
/* arrange wave w's window for (walk wl, centre pos): tokens [pos-R, pos+R] */
static void window_move(Cfg *c, Wave *w, int64_t wl, int pos, int64_t *cur_wl)
{
    const int32_t *walk = c->walks + wl * c->walk_len;
    if (*cur_wl != wl) {          /* flush the previous walk, load tokens [0, R) ... the loop below brings [.., pos+R] */
        for (int i = 0; i < w->nslots; ++i) if (w->slots[i].node >= 0) slot_writeback(c, &w->slots[i]);
        *cur_wl = wl; w->win_pos = -1;
        for (int q = 0; q < c->R && q < c->walk_len; ++q) slot_enter(c, w, walk[q]);
    }
    while (w->win_pos < pos) {
        ++w->win_pos;
        if (w->win_pos + c->R < c->walk_len) slot_enter(c, w, walk[w->win_pos + c->R]);
        if (w->win_pos - c->R - 1 >= 0) slot_leave(c, w, walk[w->win_pos - c->R - 1]);   /* token p-R leaves AFTER centre p: done when p+1 is arranged */
    }
}

static void load_negs(Cfg *c, Pair *p)
{
    for (int j = 0; j < NEG; ++j) {
        memcpy(p->ny + (size_t)j * c->d, c->SynNeg + (size_t)p->tgt[j] * c->d, sizeof(float) * c->d);
        memcpy(p->nl + (size_t)j * c->d, p->ny + (size_t)j * c->d, sizeof(float) * c->d);
    }
    p->loaded = 1;
}

static float grad(float f, float label, float alpha)
{
    if (f > 6.0f) return (label - 1.0f) * alpha;
    if (f < -6.0f) return label * alpha;
    return (label - 1.0f + 1.0f / (1.0f + expf(f))) * alpha;
}

void hogwild_emul_train(int64_t n, int32_t d, int64_t nwalks, int32_t walk_len, const int32_t *walks, int32_t window, float alpha0,
                        int32_t epochs, int32_t epoch, int64_t tokens_total, int64_t token_offset, int64_t walk_id_offset,
                        const float *UT, const int32_t *KT, uint64_t seed, int32_t flags, float *SynPos, float *SynNeg,
                        int32_t W, int32_t L, int32_t R, int32_t ctr_mode, int32_t ctx_mode, int32_t neg_mode, int64_t *stats,
                        const int32_t *counts, int32_t hot_thr)
{
    Cfg c = {n, d, nwalks, walk_len, walks, window, alpha0, (int64_t)epochs * tokens_total + 1, token_offset, walk_id_offset, epoch, UT, KT,
             seed, flags, SynPos, SynNeg, W, L, R, ctr_mode, ctx_mode, neg_mode, counts, counts ? hot_thr : 0, 0, 0, 0};
    if (L > MAXL) L = c.L = MAXL;
    Wave *wv = (Wave *)calloc((size_t)W, sizeof(Wave));
    int64_t *cur_wl = (int64_t *)malloc(sizeof(int64_t) * (size_t)W);
    const int nslots = 2 * R + 2;
    for (int g = 0; g < W; ++g) {
        Wave *w = &wv[g];
        w->wl = g; w->pos = 0; w->walk_live = g < nwalks; w->qh = 0; w->qn = 0; w->cword = -1; cur_wl[g] = -1;
        w->pool = (float *)malloc(sizeof(float) * (size_t)d * ((size_t)MAXQ * 2 * NEG + 2 + 2 * (size_t)nslots));
        float *p = w->pool;
        for (int i = 0; i < MAXQ; ++i) { w->q[i].ny = p; p += (size_t)NEG * d; w->q[i].nl = p; p += (size_t)NEG * d; }
        w->yp = p; p += d; w->ypl = p; p += d;
        w->slots = (Slot *)calloc((size_t)nslots, sizeof(Slot)); w->nslots = nslots;
        for (int i = 0; i < nslots; ++i) { w->slots[i].node = -1; w->slots[i].work = p; p += d; w->slots[i].orig = p; p += d; }
    }
    float *neu = (float *)malloc(sizeof(float) * (size_t)d);
    int live = W;
    while (live > 0) {
        live = 0;
        for (int g = 0; g < W; ++g) {
            Wave *w = &wv[g];
            while (w->qn < L + 1 && w->qn + 2 * window <= MAXQ && gen_centre(&c, w)) {}
            if (w->qn == 0) {
                if (ctx_mode && cur_wl[g] >= 0) {      /* flush the window of the last walk */
                    for (int i = 0; i < w->nslots; ++i) if (w->slots[i].node >= 0) slot_writeback(&c, &w->slots[i]);
                    cur_wl[g] = -1;
                }
                continue;
            }
            ++live;
            Pair *p = &w->q[w->qh];
            /* prefetch: the negative rows of the pairs up to L ahead are read NOW (kernel: issue(P2) at the top of step) */
            if (neg_mode)
                for (int k = 0; k <= L && k < w->qn; ++k) {          /* ... within the centre: the kernel's prefetch does not cross a centre boundary */
                    Pair *f = &w->q[(w->qh + k) % MAXQ];
                    if (!f->loaded) load_negs(&c, f);
                    if (f->last) break;
                }
            ++c.stat_pairs;
            /* --- centre row */
            float *yp;
            if (ctr_mode == 0) yp = SynNeg + (size_t)p->word * d;
            else {
                if (p->first) { memcpy(w->yp, SynNeg + (size_t)p->word * d, sizeof(float) * d); memcpy(w->ypl, w->yp, sizeof(float) * d); w->cword = p->word; }
                yp = w->yp;
            }
            /* --- context row */
            float *xc;
            if (ctx_mode == 0) xc = SynPos + (size_t)p->ctx * d;
            else {
                if (p->first) window_move(&c, w, p->wl, p->pos, &cur_wl[g]);
                Slot *s = is_hot(&c, p->ctx) ? NULL : slot_find(w, p->ctx);
                xc = s ? s->work : SynPos + (size_t)p->ctx * d;        /* a hot row: fetched for the pair, updated in place (atomic add of neu1e) */
            }
            for (int k = 0; k < d; ++k) neu[k] = 0.0f;
            {   /* positive target */
                float f = 0.0f;
                for (int k = 0; k < d; ++k) f += xc[k] * yp[k];
                const float gg = grad(f, 1.0f, p->alpha);
                for (int k = 0; k < d; ++k) { neu[k] += gg * yp[k]; yp[k] += gg * xc[k]; }
            }
            for (int j = 0; j < NEG; ++j) {
                const int32_t tg = p->tgt[j];
                if (tg == p->word) continue;
                float *gy = SynNeg + (size_t)tg * d;
                if (neg_mode == 0) {
                    float f = 0.0f;
                    for (int k = 0; k < d; ++k) f += xc[k] * gy[k];
                    const float gg = grad(f, 0.0f, p->alpha);
                    for (int k = 0; k < d; ++k) { neu[k] += gg * gy[k]; gy[k] += gg * xc[k]; }
                } else {
                    float *y = p->ny + (size_t)j * d, *yl = p->nl + (size_t)j * d;
                    /* the kernel re-fetches rows it updated itself since the prefetch (special mask): same target in an earlier slot of this
                     * pair or in the previous L pairs of this wave -> take the current shared value (the wave's own store is in it) */
                    int own = 0;
                    for (int jp = 0; jp < j; ++jp) own |= p->tgt[jp] == tg;
                    for (int k = 1; k <= L && !own; ++k) {
                        const Pair *e = &w->q[(w->qh + MAXQ - k) % MAXQ];       /* executed pairs stay readable in the ring */
                        for (int jp = 0; jp < NEG; ++jp) own |= e->tgt[jp] == tg;
                    }
                    if (own) { memcpy(y, gy, sizeof(float) * d); memcpy(yl, gy, sizeof(float) * d); }
                    if (memcmp(yl, gy, sizeof(float) * d) != 0 && !(neg_mode == 1 && is_hot(&c, tg))) ++c.stat_neg_lost;      /* someone else stored this row since we read it */
                    float f = 0.0f;
                    for (int k = 0; k < d; ++k) f += xc[k] * y[k];
                    const float gg = grad(f, 0.0f, p->alpha);
                    for (int k = 0; k < d; ++k) neu[k] += gg * y[k];
                    if (neg_mode == 1 && !is_hot(&c, tg)) for (int k = 0; k < d; ++k) gy[k] = y[k] + gg * xc[k];
                    else                                  for (int k = 0; k < d; ++k) gy[k] = gy[k] + gg * xc[k];     /* delta at store time / hot row: atomic add */
                }
            }
            for (int k = 0; k < d; ++k) xc[k] += neu[k];
            if (ctr_mode && p->last) {
                float *gy = SynNeg + (size_t)p->word * d;
                if (memcmp(w->ypl, gy, sizeof(float) * d) != 0) ++c.stat_ctr_lost;
                if (ctr_mode == 1) memcpy(gy, w->yp, sizeof(float) * d);
                else for (int k = 0; k < d; ++k) gy[k] = gy[k] + (w->yp[k] - w->ypl[k]);
            }
            p->loaded = 0;
            w->qh = (w->qh + 1) % MAXQ; --w->qn;
        }
    }
    if (stats) { stats[0] = c.stat_pairs; stats[1] = c.stat_neg_lost; stats[2] = c.stat_ctr_lost; }
    free(neu); free(cur_wl);
    for (int g = 0; g < W; ++g) { free(wv[g].pool); free(wv[g].slots); }
    free(wv);
}
