/*
 * popoa_oracle.c — TEST INFRASTRUCTURE ONLY (see cl_oracle.h: parity PINNED against oracle/_ref and
 * tests/golden).  A deliberately literal, single-threaded C restatement of the reference's
 * between-anchor alignment path.  Every function cites the reference lines it follows
 * (paths relative to the reference root).  Not used, linked or imported by the product.
 */
#include "cl_oracle.h"

#include <limits.h>
#include <stdlib.h>
#include <string.h>

#define MININF (INT32_MIN / 2) /* cell_t::mininf, include/centrolign/alignment.hpp:740 */

/* ---------------------------------------------------------------------------------------------------- */
/* small graph helpers                                                                                  */

typedef struct {
    uint64_t* off;
    uint32_t* idx;
    int owned;
} adj_t;

/* BaseGraph keeps both next and prev lists (include/centrolign/graph.hpp:139-147).  When the caller gave
 * only prev lists we rebuild next lists by scanning nodes in id order; only the ORDER inside a next list
 * can differ from the reference's, and nothing below depends on it (topological ties do not change DP
 * values or the traceback, which reads previous() only). */
static int get_next(const clo_graph* g, adj_t* out) {
    if (g->next_off) {
        out->off = (uint64_t*)g->next_off;
        out->idx = (uint32_t*)g->next_idx;
        out->owned = 0;
        return 0;
    }
    uint64_t n = g->n;
    uint64_t base = n ? g->prev_off[0] : 0;
    uint64_t e = n ? g->prev_off[n] - base : 0;
    out->off = (uint64_t*)calloc(n + 2, sizeof(uint64_t));
    out->idx = (uint32_t*)malloc((e ? e : 1) * sizeof(uint32_t));
    out->owned = 1;
    if (!out->off || !out->idx) return CL_ERR_OUT_OF_MEMORY;
    for (uint64_t v = 0; v < n; ++v)
        for (uint64_t k = g->prev_off[v]; k < g->prev_off[v + 1]; ++k) out->off[g->prev_idx[k] + 2]++;
    for (uint64_t v = 0; v < n; ++v) out->off[v + 2] += out->off[v + 1];
    for (uint64_t v = 0; v < n; ++v)
        for (uint64_t k = g->prev_off[v]; k < g->prev_off[v + 1]; ++k) out->idx[out->off[g->prev_idx[k] + 1]++] = (uint32_t)v;
    return 0;
}

static void free_adj(adj_t* a) {
    if (a->owned) {
        free(a->off);
        free(a->idx);
    }
}

/* Kahn's algorithm with a LIFO stack seeded in ascending id order
 * (include/centrolign/topological_order.hpp:12-60). returns malloc'ed order or NULL (cycle / OOM). */
static uint32_t* topological_order(const clo_graph* g, const adj_t* next) {
    uint64_t n = g->n;
    uint32_t* order = (uint32_t*)malloc((n ? n : 1) * sizeof(uint32_t));
    uint32_t* stack = (uint32_t*)malloc((n ? n : 1) * sizeof(uint32_t));
    uint64_t* indeg = (uint64_t*)malloc((n ? n : 1) * sizeof(uint64_t));
    if (!order || !stack || !indeg) {
        free(order); free(stack); free(indeg);
        return NULL;
    }
    uint64_t sp = 0, no = 0;
    for (uint64_t v = 0; v < n; ++v) {
        indeg[v] = g->prev_off[v + 1] - g->prev_off[v];
        if (indeg[v] == 0) stack[sp++] = (uint32_t)v;
    }
    while (sp) {
        uint32_t v = stack[--sp];
        order[no++] = v;
        for (uint64_t k = next->off[v]; k < next->off[v + 1]; ++k) {
            uint32_t w = next->idx[k];
            if (--indeg[w] == 0) stack[sp++] = w;
        }
    }
    free(stack);
    free(indeg);
    if (no != n) {
        free(order);
        return NULL;
    }
    return order;
}

/* ---------------------------------------------------------------------------------------------------- */
/* po_poa_internal<true, NumPW>  (include/centrolign/alignment.hpp:753-1151)                            */

static inline int32_t imax(int32_t a, int32_t b) { return a > b ? a : b; }

int clo_po_poa(const clo_graph* g1, const clo_graph* g2, int npw, const cl_align_params* prm,
               uint64_t* pairs_out, uint64_t* n_pairs_out, int64_t* score_out) {
    const uint64_t n1 = g1->n, n2 = g2->n;
    const uint64_t W = n2 + 1;
    const int S = 1 + 2 * npw; /* ints per cell: M, I[npw], D[npw]  (cell_t, alignment.hpp:738-751) */
    int rc = 0;
    adj_t nx1 = {0, 0, 0}, nx2 = {0, 0, 0};
    uint32_t *order1 = NULL, *order2 = NULL;
    uint8_t *is_src1 = NULL, *is_src2 = NULL;
    int32_t* dp = NULL;

    if ((rc = get_next(g1, &nx1)) || (rc = get_next(g2, &nx2))) goto done;
    order1 = topological_order(g1, &nx1); /* alignment.hpp:806-807 */
    order2 = topological_order(g2, &nx2);
    if (!order1 || !order2) { rc = CL_ERR_CYCLIC_GRAPH; goto done; }

    /* alignment.hpp:789-790: (n1+1) x (n2+1) cells, last row/column = boundary */
    dp = (int32_t*)malloc((n1 + 1) * W * (uint64_t)S * sizeof(int32_t));
    is_src1 = (uint8_t*)calloc(n1 + 1, 1);
    is_src2 = (uint8_t*)calloc(n2 + 1, 1);
    if (!dp || !is_src1 || !is_src2) { rc = CL_ERR_OUT_OF_MEMORY; goto done; }
    for (uint64_t c = 0; c < (n1 + 1) * W * (uint64_t)S; ++c) dp[c] = MININF;

#define CELL(i, j) (dp + ((uint64_t)(i) * W + (uint64_t)(j)) * (uint64_t)S)
#define cM(c) ((c)[0])
#define cI(c, k) ((c)[1 + (k)])
#define cD(c, k) ((c)[1 + npw + (k)])
#define SCORE(i, j) ((int32_t)(g1->label[i] == g2->label[j] ? prm->match : (uint32_t)(-(int32_t)prm->mismatch)))

    int32_t oe[3], ex[3];
    for (int k = 0; k < npw; ++k) {
        oe[k] = (int32_t)(prm->gap_open[k] + prm->gap_extend[k]);
        ex[k] = (int32_t)prm->gap_extend[k];
    }

    /* alignment.hpp:813-829: boundary initialisation */
    for (uint64_t a = 0; a < g1->n_src; ++a) {
        uint32_t s1 = g1->src[a];
        for (uint64_t b = 0; b < g2->n_src; ++b) cM(CELL(s1, g2->src[b])) = SCORE(s1, g2->src[b]);
        for (int k = 0; k < npw; ++k) cI(CELL(s1, n2), k) = -oe[k];
    }
    for (uint64_t b = 0; b < g2->n_src; ++b)
        for (int k = 0; k < npw; ++k) cD(CELL(n1, g2->src[b]), k) = -oe[k];

    /* alignment.hpp:832-862: DP along initial insertions (boundary column n2) */
    for (uint64_t t = 0; t < n1; ++t) {
        uint32_t i = order1[t];
        int32_t* cell = CELL(i, n2);
        for (int k = 0; k < npw; ++k) cM(cell) = imax(cM(cell), cI(cell, k));
        for (uint64_t e = nx1.off[i]; e < nx1.off[i + 1]; ++e) {
            int32_t* nc = CELL(nx1.idx[e], n2);
            for (int k = 0; k < npw; ++k) cI(nc, k) = imax(cI(nc, k), cI(cell, k) - ex[k]);
        }
        for (uint64_t b = 0; b < g2->n_src; ++b) {
            int32_t* nc = CELL(i, g2->src[b]);
            for (int k = 0; k < npw; ++k) cD(nc, k) = imax(cD(nc, k), cM(cell) - oe[k]);
        }
        for (uint64_t e = nx1.off[i]; e < nx1.off[i + 1]; ++e) {
            uint32_t ni = nx1.idx[e];
            for (uint64_t b = 0; b < g2->n_src; ++b) {
                int32_t* nc = CELL(ni, g2->src[b]);
                cM(nc) = imax(cM(nc), cM(cell) + SCORE(ni, g2->src[b]));
            }
        }
    }
    /* alignment.hpp:864-894: DP along initial deletions (boundary row n1) */
    for (uint64_t t = 0; t < n2; ++t) {
        uint32_t j = order2[t];
        int32_t* cell = CELL(n1, j);
        for (int k = 0; k < npw; ++k) cM(cell) = imax(cM(cell), cD(cell, k));
        for (uint64_t e = nx2.off[j]; e < nx2.off[j + 1]; ++e) {
            int32_t* nc = CELL(n1, nx2.idx[e]);
            for (int k = 0; k < npw; ++k) cD(nc, k) = imax(cD(nc, k), cD(cell, k) - ex[k]);
        }
        for (uint64_t a = 0; a < g1->n_src; ++a) {
            int32_t* nc = CELL(g1->src[a], j);
            for (int k = 0; k < npw; ++k) cI(nc, k) = imax(cI(nc, k), cM(cell) - oe[k]);
        }
        for (uint64_t e = nx2.off[j]; e < nx2.off[j + 1]; ++e) {
            uint32_t nj = nx2.idx[e];
            for (uint64_t a = 0; a < g1->n_src; ++a) {
                int32_t* nc = CELL(g1->src[a], nj);
                cM(nc) = imax(cM(nc), cM(cell) + SCORE(g1->src[a], nj));
            }
        }
    }
    /* alignment.hpp:897-938: interior */
    for (uint64_t t = 0; t < n1; ++t) {
        uint32_t i = order1[t];
        for (uint64_t u = 0; u < n2; ++u) {
            uint32_t j = order2[u];
            int32_t* cell = CELL(i, j);
            for (int k = 0; k < npw; ++k) cM(cell) = imax(cM(cell), imax(cI(cell, k), cD(cell, k)));
            for (uint64_t e = nx1.off[i]; e < nx1.off[i + 1]; ++e) {
                int32_t* nc = CELL(nx1.idx[e], j);
                for (int k = 0; k < npw; ++k)
                    cI(nc, k) = imax(cI(nc, k), imax(cM(cell) - oe[k], cI(cell, k) - ex[k]));
            }
            for (uint64_t e = nx2.off[j]; e < nx2.off[j + 1]; ++e) {
                int32_t* nc = CELL(i, nx2.idx[e]);
                for (int k = 0; k < npw; ++k)
                    cD(nc, k) = imax(cD(nc, k), imax(cM(cell) - oe[k], cD(cell, k) - ex[k]));
            }
            for (uint64_t e = nx1.off[i]; e < nx1.off[i + 1]; ++e) {
                uint32_t ni = nx1.idx[e];
                for (uint64_t f = nx2.off[j]; f < nx2.off[j + 1]; ++f) {
                    uint32_t nj = nx2.idx[f];
                    int32_t* nc = CELL(ni, nj);
                    cM(nc) = imax(cM(nc), cM(cell) + SCORE(ni, nj));
                }
            }
        }
    }

    /* alignment.hpp:979-1008: best end cell; first strictly better in the given sink order */
    uint64_t tb1 = UINT64_MAX, tb2 = UINT64_MAX;
    if (n1 != 0 && n2 != 0) {
        for (uint64_t a = 0; a < g1->n_snk; ++a)
            for (uint64_t b = 0; b < g2->n_snk; ++b)
                if (tb1 == UINT64_MAX || cM(CELL(g1->snk[a], g2->snk[b])) > cM(CELL(tb1, tb2))) {
                    tb1 = g1->snk[a];
                    tb2 = g2->snk[b];
                }
    } else if (n1 != 0) {
        for (uint64_t a = 0; a < g1->n_snk; ++a)
            if (tb1 == UINT64_MAX || cM(CELL(g1->snk[a], 0)) > cM(CELL(tb1, 0))) {
                tb1 = g1->snk[a];
                tb2 = 0;
            }
    } else if (n2 != 0) {
        for (uint64_t b = 0; b < g2->n_snk; ++b)
            if (tb2 == UINT64_MAX || cM(CELL(0, g2->snk[b])) > cM(CELL(0, tb2))) {
                tb1 = 0;
                tb2 = g2->snk[b];
            }
    }
    if (score_out) *score_out = (tb1 != UINT64_MAX) ? (int64_t)cM(CELL(tb1, tb2)) : 0; /* :1018-1025 */

    for (uint64_t a = 0; a < g1->n_src; ++a) is_src1[g1->src[a]] = 1; /* :1027-1029 */
    for (uint64_t b = 0; b < g2->n_src; ++b) is_src2[g2->src[b]] = 1;

    /* alignment.hpp:1036-1138: traceback */
    uint64_t np = 0;
    int comp = 0;
    while (tb1 != UINT64_MAX && tb2 != UINT64_MAX) {
        uint64_t h1 = tb1, h2 = tb2;
        tb1 = UINT64_MAX;
        tb2 = UINT64_MAX;
        const int32_t* cell = CELL(h1, h2);
        if (comp == 0) {
            for (int k = 0; k < npw; ++k) {
                if (cM(cell) == cI(cell, k)) { comp = k + 1; break; }
                if (cM(cell) == cD(cell, k)) { comp = -k - 1; break; }
            }
        }
        /* predecessor lists: graph edges unless in the boundary, then the boundary index if a source */
        uint64_t p1b = 0, p1e = 0, p2b = 0, p2e = 0;
        if (h1 < n1) { p1b = g1->prev_off[h1]; p1e = g1->prev_off[h1 + 1]; }
        if (h2 < n2) { p2b = g2->prev_off[h2]; p2e = g2->prev_off[h2 + 1]; }
        int x1 = (h1 < n1 && is_src1[h1]) ? 1 : 0; /* sources1_set.count(n1) is never true */
        int x2 = (h2 < n2 && is_src2[h2]) ? 1 : 0;
#define P1(e) ((e) < p1e ? (uint64_t)g1->prev_idx[e] : n1)
#define P2(e) ((e) < p2e ? (uint64_t)g2->prev_idx[e] : n2)
        if (comp == 0) {
            pairs_out[2 * np] = h1;
            pairs_out[2 * np + 1] = h2;
            ++np;
            int32_t sc = SCORE(h1, h2);
            for (uint64_t e = p1b; e < p1e + x1; ++e) {
                for (uint64_t f = p2b; f < p2e + x2; ++f) {
                    if (cM(CELL(P1(e), P2(f))) + sc == cM(cell)) {
                        tb1 = P1(e);
                        tb2 = P2(f);
                        break; /* leaves the inner loop only: the last prev1 with a hit wins */
                    }
                }
            }
        } else if (comp > 0) {
            pairs_out[2 * np] = h1;
            pairs_out[2 * np + 1] = CL_GAP;
            ++np;
            int k = comp - 1;
            for (uint64_t e = p1b; e < p1e + x1; ++e) {
                const int32_t* pc = CELL(P1(e), h2);
                if (cI(cell, k) == cM(pc) - oe[k]) { comp = 0; tb1 = P1(e); tb2 = h2; break; }
                if (cI(cell, k) == cI(pc, k) - ex[k]) { tb1 = P1(e); tb2 = h2; break; }
            }
        } else {
            pairs_out[2 * np] = CL_GAP;
            pairs_out[2 * np + 1] = h2;
            ++np;
            int k = -comp - 1;
            for (uint64_t f = p2b; f < p2e + x2; ++f) {
                const int32_t* pc = CELL(h1, P2(f));
                if (cD(cell, k) == cM(pc) - oe[k]) { comp = 0; tb1 = h1; tb2 = P2(f); break; }
                if (cD(cell, k) == cD(pc, k) - ex[k]) { tb1 = h1; tb2 = P2(f); break; }
            }
        }
    }
    /* alignment.hpp:1141 */
    for (uint64_t a = 0, b = np; a + 1 < b; ++a) {
        --b;
        uint64_t t0 = pairs_out[2 * a], t1 = pairs_out[2 * a + 1];
        pairs_out[2 * a] = pairs_out[2 * b];
        pairs_out[2 * a + 1] = pairs_out[2 * b + 1];
        pairs_out[2 * b] = t0;
        pairs_out[2 * b + 1] = t1;
    }
    *n_pairs_out = np;

done:
    free(dp);
    free(is_src1);
    free(is_src2);
    free(order1);
    free(order2);
    free_adj(&nx1);
    free_adj(&nx2);
    return rc;
}

/* ---------------------------------------------------------------------------------------------------- */
/* pure_deletion_alignment (alignment.hpp:1178-1210) over shortest_path (shortest_path.hpp:32-100)       */

int clo_pure_deletion(const clo_graph* g, int npw, const cl_align_params* prm, uint64_t* pairs_out,
                      uint64_t* n_pairs_out, int64_t* score_out) {
    uint64_t n = g->n;
    *n_pairs_out = 0;
    if (score_out) *score_out = 0;
    if (n == 0) return 0;
    adj_t nx = {0, 0, 0};
    int rc = get_next(g, &nx);
    if (rc) return rc;
    uint32_t* order = topological_order(g, &nx);
    uint64_t* dp = (uint64_t*)malloc(n * sizeof(uint64_t));
    if (!order || !dp) {
        free(order); free(dp); free_adj(&nx);
        return order ? CL_ERR_OUT_OF_MEMORY : CL_ERR_CYCLIC_GRAPH;
    }
    const uint64_t INF = (uint64_t)INT64_MAX; /* shortest_path.hpp:58 */
    for (uint64_t v = 0; v < n; ++v) dp[v] = INF;
    for (uint64_t a = 0; a < g->n_src; ++a) dp[g->src[a]] = 0;
    for (uint64_t t = 0; t < n; ++t) {
        uint32_t v = order[t];
        uint64_t thru = dp[v] + 1; /* label_size == 1 for a BaseGraph; INF+1 stays "large", as in the reference */
        for (uint64_t e = nx.off[v]; e < nx.off[v + 1]; ++e)
            if (thru < dp[nx.idx[e]]) dp[nx.idx[e]] = thru;
    }
    uint64_t best = UINT64_MAX;
    for (uint64_t b = 0; b < g->n_snk; ++b) {
        uint32_t v = g->snk[b];
        if (dp[v] != INF && (best == UINT64_MAX || dp[v] < dp[best])) best = v; /* shortest_path.hpp:78-84 */
    }
    uint64_t np = 0;
    if (best != UINT64_MAX) {
        uint64_t cur = best;
        pairs_out[2 * np] = cur; pairs_out[2 * np + 1] = CL_GAP; ++np;
        while (dp[cur] != 0) {
            uint64_t nxt = UINT64_MAX;
            for (uint64_t e = g->prev_off[cur]; e < g->prev_off[cur + 1]; ++e)
                if (dp[g->prev_idx[e]] + 1 == dp[cur]) { nxt = g->prev_idx[e]; break; }
            if (nxt == UINT64_MAX) break; /* the reference would spin forever; unreachable for valid input */
            cur = nxt;
            pairs_out[2 * np] = cur; pairs_out[2 * np + 1] = CL_GAP; ++np;
        }
        for (uint64_t a = 0, b = np; a + 1 < b; ++a) {
            --b;
            uint64_t t0 = pairs_out[2 * a];
            pairs_out[2 * a] = pairs_out[2 * b];
            pairs_out[2 * b] = t0;
        }
    }
    *n_pairs_out = np;
    if (score_out) {
        if (np == 0) *score_out = 0;
        else {
            /* alignment.hpp:1202-1205 literally: min over pw of (-open - extend), evaluated in uint32 then widened.
             * (the value is never consumed by the stitcher; kept for parity of the score_out argument) */
            int64_t s = INT64_MAX;
            for (int k = 0; k < npw; ++k) {
                uint32_t u = (uint32_t)(0u - prm->gap_open[k] - prm->gap_extend[k]);
                if ((int64_t)u < s) s = (int64_t)u;
            }
            *score_out = s;
        }
    }
    free(order); free(dp); free_adj(&nx);
    return 0;
}

/* Extractor::source_sink_minmax (src/anchorer.cpp:14-23) over minmax_distance (minmax_distance.hpp:16-72) */
int clo_source_sink_minmax(const clo_graph* g, int64_t* min_out, int64_t* max_out) {
    uint64_t n = g->n;
    *min_out = INT64_MAX;
    *max_out = -1;
    if (n == 0) return 0;
    adj_t nx = {0, 0, 0};
    int rc = get_next(g, &nx);
    if (rc) return rc;
    uint32_t* order = topological_order(g, &nx);
    int64_t* mn = (int64_t*)malloc(n * sizeof(int64_t));
    int64_t* mx = (int64_t*)malloc(n * sizeof(int64_t));
    if (!order || !mn || !mx) {
        free(order); free(mn); free(mx); free_adj(&nx);
        return order ? CL_ERR_OUT_OF_MEMORY : CL_ERR_CYCLIC_GRAPH;
    }
    for (uint64_t v = 0; v < n; ++v) { mn[v] = INT64_MAX; mx[v] = -1; }
    for (uint64_t a = 0; a < g->n_src; ++a) { mn[g->src[a]] = 0; mx[g->src[a]] = 0; }
    for (uint64_t t = 0; t < n; ++t) {
        uint32_t v = order[t];
        if (mn[v] != INT64_MAX)
            for (uint64_t e = nx.off[v]; e < nx.off[v + 1]; ++e) {
                uint32_t w = nx.idx[e];
                if (mn[v] + 1 < mn[w]) mn[w] = mn[v] + 1;
                if (mx[v] + 1 > mx[w]) mx[w] = mx[v] + 1;
            }
    }
    for (uint64_t b = 0; b < g->n_snk; ++b) {
        if (mn[g->snk[b]] < *min_out) *min_out = mn[g->snk[b]];
        if (mx[g->snk[b]] > *max_out) *max_out = mx[g->snk[b]];
    }
    free(order); free(mn); free(mx); free_adj(&nx);
    return 0;
}

/* src/stitcher.cpp:31-52 */
int clo_choose_num_pw(uint64_t n1, uint64_t n2, const cl_align_params* p) {
    uint64_t cutoffs[2];
    for (int i = 1; i < 3; ++i) {
        if (p->gap_open[i - 1] > p->gap_open[i] || p->gap_extend[i - 1] < p->gap_extend[i]) return CL_ERR_BAD_GAP_PARAMS;
        uint32_t diff_open = p->gap_open[i] - p->gap_open[i - 1];
        uint32_t diff_extend = p->gap_extend[i - 1] - p->gap_extend[i];
        if (diff_extend == 0) return CL_ERR_BAD_GAP_PARAMS; /* the reference divides by zero here */
        cutoffs[i - 1] = (diff_open + diff_extend - 1) / diff_extend;
    }
    int c = 0;
    while (c < 2 && n1 > cutoffs[c] && n2 > cutoffs[c]) ++c;
    return c + 1;
}

/* include/centrolign/stitcher.hpp:268-360 */
int clo_route(const clo_graph* g1, const clo_graph* g2, int only_del, const cl_stitch_params* sp) {
    if (g2->n == 0) return CL_ROUTE_PURE_DELETION_1;
    if (g1->n == 0) return CL_ROUTE_PURE_DELETION_2;
    uint64_t mat = (g1->n + 1) * (g2->n + 1);
    if (mat <= sp->min_wfa_size && (!only_del || mat <= sp->max_trivial_size)) return CL_ROUTE_PO_POA;
    int64_t mn1, mx1, mn2, mx2;
    clo_source_sink_minmax(g1, &mn1, &mx1);
    clo_source_sink_minmax(g2, &mn2, &mx2);
    /* size_t arithmetic in the reference (stitcher.hpp:291-293) */
    uint64_t min1 = (uint64_t)mn1, max1 = (uint64_t)mx1, min2 = (uint64_t)mn2, max2 = (uint64_t)mx2;
    if (max1 * sp->deletion_alignment_ratio <= min2 && max1 <= sp->deletion_alignment_short_max_size &&
        min2 >= sp->deletion_alignment_long_min_size)
        return CL_ROUTE_DELETION_WFA_1;
    if (max2 * sp->deletion_alignment_ratio <= min1 && max2 <= sp->deletion_alignment_short_max_size &&
        min1 >= sp->deletion_alignment_long_min_size)
        return CL_ROUTE_DELETION_WFA_2;
    double r = sp->max_wfa_ratio;
    if (mat < sp->max_wfa_size &&
        ((min2 * r >= min1 && min2 <= max1 * r) || (max2 * r >= min1 && max2 <= max1 * r) ||
         (min1 * r >= min2 && min1 <= max2 * r) || (max1 * r >= min2 && max1 <= max2 * r)) &&
        !only_del)
        return CL_ROUTE_PWFA;
    return CL_ROUTE_GREEDY_PARTIAL;
}

/* ---------------------------------------------------------------------------------------------------- */
/* batch driver: Stitcher::subalign (src/stitcher.cpp:24-78) for every problem                           */

static void side_graph(const cl_graph_side* s, uint64_t k, clo_graph* g) {
    uint64_t b = s->node_off[k];
    g->n = s->node_off[k + 1] - b;
    g->label = s->label + b;
    g->prev_off = s->prev_off + b;
    g->prev_idx = s->prev_idx;
    g->next_off = s->next_off ? s->next_off + b : NULL;
    g->next_idx = s->next_idx;
    g->n_src = s->src_off[k + 1] - s->src_off[k];
    g->src = s->src_idx + s->src_off[k];
    g->n_snk = s->snk_off[k + 1] - s->snk_off[k];
    g->snk = s->snk_idx + s->snk_off[k];
}

void clo_result_free(cl_stitch_result* r) {
    if (!r) return;
    free(r->aln_off); free(r->pairs); free(r->score); free(r->route); free(r->num_pw);
    memset(r, 0, sizeof(*r));
}

int clo_stitch_batch(const cl_stitch_batch* batch, const cl_stitch_params* sp, const uint8_t* force_num_pw,
                     cl_stitch_result* out) {
    uint64_t n = batch->n_problems;
    memset(out, 0, sizeof(*out));
    out->n_problems = n;
    uint64_t cap = batch->side[0].node_off[n] + batch->side[1].node_off[n];
    out->aln_off = (uint64_t*)calloc(n + 1, sizeof(uint64_t));
    out->pairs = (uint64_t*)malloc((cap ? cap : 1) * 2 * sizeof(uint64_t));
    out->score = (int64_t*)calloc(n ? n : 1, sizeof(int64_t));
    out->route = (uint8_t*)calloc(n ? n : 1, 1);
    out->num_pw = (uint8_t*)calloc(n ? n : 1, 1);
    if (!out->aln_off || !out->pairs || !out->score || !out->route || !out->num_pw) {
        clo_result_free(out);
        return CL_ERR_OUT_OF_MEMORY;
    }
    uint64_t np_total = 0;
    for (uint64_t k = 0; k < n; ++k) {
        clo_graph g1, g2;
        side_graph(&batch->side[0], k, &g1);
        side_graph(&batch->side[1], k, &g2);
        int npw = force_num_pw ? force_num_pw[k] : clo_choose_num_pw(g1.n, g2.n, &sp->alignment_params);
        if (npw < 1 || npw > 3) { clo_result_free(out); return npw < 0 ? npw : CL_ERR_INVALID_ARGUMENT; }
        int only_del = batch->only_deletion_alns ? batch->only_deletion_alns[k] : 0;
        int route;
        if (force_num_pw) route = g2.n == 0 ? CL_ROUTE_PURE_DELETION_1 : g1.n == 0 ? CL_ROUTE_PURE_DELETION_2 : CL_ROUTE_PO_POA;
        else route = clo_route(&g1, &g2, only_del, sp);
        uint64_t* pairs = out->pairs + 2 * np_total;
        uint64_t np = 0;
        int rc = 0;
        switch (route) {
        case CL_ROUTE_PURE_DELETION_1:
            rc = clo_pure_deletion(&g1, npw, &sp->alignment_params, pairs, &np, &out->score[k]);
            break;
        case CL_ROUTE_PURE_DELETION_2:
            rc = clo_pure_deletion(&g2, npw, &sp->alignment_params, pairs, &np, &out->score[k]);
            for (uint64_t a = 0; a < np; ++a) { /* swap_graphs, src/alignment.cpp:41-45 */
                uint64_t t = pairs[2 * a]; pairs[2 * a] = pairs[2 * a + 1]; pairs[2 * a + 1] = t;
            }
            break;
        case CL_ROUTE_PO_POA:
            rc = clo_po_poa(&g1, &g2, npw, &sp->alignment_params, pairs, &np, &out->score[k]);
            break;
        default:
            rc = CL_ERR_UNSUPPORTED_ROUTE;
        }
        if (rc) { clo_result_free(out); return rc; }
        /* translate, src/alignment.cpp:26-39 */
        const uint64_t* bt1 = batch->side[0].back_translation;
        const uint64_t* bt2 = batch->side[1].back_translation;
        for (uint64_t a = 0; a < np; ++a) {
            if (bt1 && pairs[2 * a] != CL_GAP) pairs[2 * a] = bt1[batch->side[0].node_off[k] + pairs[2 * a]];
            if (bt2 && pairs[2 * a + 1] != CL_GAP) pairs[2 * a + 1] = bt2[batch->side[1].node_off[k] + pairs[2 * a + 1]];
        }
        out->route[k] = (uint8_t)route;
        out->num_pw[k] = (uint8_t)npw;
        np_total += np;
        out->aln_off[k + 1] = np_total;
    }
    return 0;
}

uint64_t clo_cells(const cl_stitch_batch* batch, uint64_t begin, uint64_t end) {
    uint64_t cells = 0;
    for (uint64_t k = begin; k < end && k < batch->n_problems; ++k) {
        uint64_t n1 = batch->side[0].node_off[k + 1] - batch->side[0].node_off[k];
        uint64_t n2 = batch->side[1].node_off[k + 1] - batch->side[1].node_off[k];
        if (n1 && n2) cells += (n1 + 1) * (n2 + 1);
    }
    return cells;
}
