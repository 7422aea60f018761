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
/* greedy_partial_alignment (alignment.hpp:1212-1611): the longest exact-match paths grown from the sources  */
/* and from the sinks, joined by a double deletion along shortest paths; when the two match paths overlap or */
/* cannot reach each other they are trimmed (bisection on the total trim) until they can.                    */

/* shortest_path between node sets (shortest_path.hpp:32-100); path_out holds n entries; returns its length */
static uint64_t sp_sets(const clo_graph* g, const adj_t* nx, const uint32_t* order, const uint64_t* from, uint64_t n_from,
                        const uint64_t* to, uint64_t n_to, uint64_t* dp, uint64_t* path_out) {
    const uint64_t INF = (uint64_t)INT64_MAX;
    for (uint64_t v = 0; v < g->n; ++v) dp[v] = INF;
    for (uint64_t a = 0; a < n_from; ++a) dp[from[a]] = 0;
    for (uint64_t t = 0; t < g->n; ++t) {
        uint32_t v = order[t];
        uint64_t thru = dp[v] + 1;
        for (uint64_t e = nx->off[v]; e < nx->off[v + 1]; ++e)
            if (thru < dp[nx->idx[e]]) dp[nx->idx[e]] = thru;
    }
    uint64_t best = UINT64_MAX;
    for (uint64_t b = 0; b < n_to; ++b)
        if (dp[to[b]] != INF && (best == UINT64_MAX || dp[to[b]] < dp[best])) best = to[b];
    if (best == UINT64_MAX) return 0;
    uint64_t np = 0, cur = best;
    path_out[np++] = cur;
    while (dp[cur] != 0) {
        uint64_t nxt = UINT64_MAX;
        for (uint64_t e = g->prev_off[cur]; e < g->prev_off[cur + 1]; ++e)
            if (dp[g->prev_idx[e]] + 1 == dp[cur]) { nxt = g->prev_idx[e]; break; }
        if (nxt == UINT64_MAX) break;
        cur = nxt;
        path_out[np++] = cur;
    }
    for (uint64_t a = 0, b = np; a + 1 < b; ++a) { --b; uint64_t t0 = path_out[a]; path_out[a] = path_out[b]; path_out[b] = t0; }
    return np;
}

/* one direction of the greedy match search (alignment.hpp:1245-1312): DFS over label-matching node pairs; the deepest
 * pair met first is the end of the path.  aln_out gets (n1, n2) pairs in the order the reference leaves them. */
static uint64_t greedy_direction(const clo_graph* g1, const clo_graph* g2, const adj_t* nx1, const adj_t* nx2, int forward,
                                 int64_t* back /* [n1*n2], -2 = unvisited */, uint64_t* stack /* [3*n1*n2 bound] */, uint64_t* aln_out) {
    const uint64_t n2 = g2->n;
    for (uint64_t i = 0; i < g1->n * n2; ++i) back[i] = -2;
    uint64_t sp = 0, max_len = 0;
    int64_t path_end = -1;
    const uint32_t* ends1 = forward ? g1->src : g1->snk;
    const uint32_t* ends2 = forward ? g2->src : g2->snk;
    const uint64_t ne1 = forward ? g1->n_src : g1->n_snk, ne2 = forward ? g2->n_src : g2->n_snk;
    for (uint64_t a = 0; a < ne1; ++a)
        for (uint64_t b = 0; b < ne2; ++b)
            if (g1->label[ends1[a]] == g2->label[ends2[b]]) {
                stack[3 * sp] = ends1[a]; stack[3 * sp + 1] = ends2[b]; stack[3 * sp + 2] = 1; ++sp;
                back[(uint64_t)ends1[a] * n2 + ends2[b]] = -1;
            }
    while (sp) {
        --sp;
        const uint64_t v1 = stack[3 * sp], v2 = stack[3 * sp + 1], len = stack[3 * sp + 2];
        if (len > max_len) { max_len = len; path_end = (int64_t)(v1 * n2 + v2); }
        const uint64_t* o1 = forward ? nx1->off : g1->prev_off;
        const uint32_t* i1 = forward ? nx1->idx : g1->prev_idx;
        const uint64_t* o2 = forward ? nx2->off : g2->prev_off;
        const uint32_t* i2 = forward ? nx2->idx : g2->prev_idx;
        for (uint64_t e1 = o1[v1]; e1 < o1[v1 + 1]; ++e1)
            for (uint64_t e2 = o2[v2]; e2 < o2[v2 + 1]; ++e2) {
                const uint64_t w1 = i1[e1], w2 = i2[e2];
                if (g1->label[w1] == g2->label[w2] && back[w1 * n2 + w2] == -2) {
                    back[w1 * n2 + w2] = (int64_t)(v1 * n2 + v2);
                    stack[3 * sp] = w1; stack[3 * sp + 1] = w2; stack[3 * sp + 2] = len + 1; ++sp;
                }
            }
    }
    uint64_t n = 0;
    while (path_end != -1) {
        aln_out[2 * n] = (uint64_t)path_end / n2; aln_out[2 * n + 1] = (uint64_t)path_end % n2; ++n;
        path_end = back[path_end];
    }
    if (forward)
        for (uint64_t a = 0, b = n; a + 1 < b; ++a) {
            --b;
            uint64_t t0 = aln_out[2 * a], t1 = aln_out[2 * a + 1];
            aln_out[2 * a] = aln_out[2 * b]; aln_out[2 * a + 1] = aln_out[2 * b + 1];
            aln_out[2 * b] = t0; aln_out[2 * b + 1] = t1;
        }
    return n;
}

int clo_greedy_partial_alignment(const clo_graph* g1, const clo_graph* g2, uint64_t* pairs_out, uint64_t* n_pairs_out) {
    *n_pairs_out = 0;
    const uint64_t n1 = g1->n, n2 = g2->n;
    adj_t nx1 = {0, 0, 0}, nx2 = {0, 0, 0};
    int rc = get_next(g1, &nx1);
    if (!rc) rc = get_next(g2, &nx2);
    if (rc) return rc;
    uint32_t* ord1 = topological_order(g1, &nx1);
    uint32_t* ord2 = topological_order(g2, &nx2);
    int64_t* back = (int64_t*)malloc((n1 * n2 ? n1 * n2 : 1) * sizeof(int64_t));
    uint64_t* stack = (uint64_t*)malloc((n1 * n2 ? n1 * n2 : 1) * 3 * sizeof(uint64_t));
    uint64_t* fwd = (uint64_t*)malloc((n1 + n2 + 1) * 2 * sizeof(uint64_t));
    uint64_t* rev = (uint64_t*)malloc((n1 + n2 + 1) * 2 * sizeof(uint64_t));
    uint64_t* dp = (uint64_t*)malloc(((n1 > n2 ? n1 : n2) + 1) * sizeof(uint64_t));
    uint64_t* p1 = (uint64_t*)malloc((n1 + 1) * sizeof(uint64_t));
    uint64_t* p2 = (uint64_t*)malloc((n2 + 1) * sizeof(uint64_t));
    uint64_t* tmp = (uint64_t*)malloc(((n1 > n2 ? n1 : n2) + 1) * sizeof(uint64_t));
    uint64_t *s1 = (uint64_t*)malloc((g1->n_src + 1) * 8), *s2 = (uint64_t*)malloc((g2->n_src + 1) * 8);
    uint64_t *k1 = (uint64_t*)malloc((g1->n_snk + 1) * 8), *k2 = (uint64_t*)malloc((g2->n_snk + 1) * 8);
    if (!ord1 || !ord2 || !back || !stack || !fwd || !rev || !dp || !p1 || !p2 || !tmp || !s1 || !s2 || !k1 || !k2) { rc = CL_ERR_OUT_OF_MEMORY; goto done; }
    for (uint64_t a = 0; a < g1->n_src; ++a) s1[a] = g1->src[a];
    for (uint64_t a = 0; a < g2->n_src; ++a) s2[a] = g2->src[a];
    for (uint64_t a = 0; a < g1->n_snk; ++a) k1[a] = g1->snk[a];
    for (uint64_t a = 0; a < g2->n_snk; ++a) k2[a] = g2->snk[a];
    {
        const uint64_t nf = greedy_direction(g1, g2, &nx1, &nx2, 1, back, stack, fwd);
        const uint64_t nr = greedy_direction(g1, g2, &nx1, &nx2, 0, back, stack, rev);
        uint64_t left_trim = 0, right_trim = 0, np1 = 0, np2 = 0;
        int found = 0;
        if (nf == 0 || nr == 0 || (fwd[2 * (nf - 1)] != rev[0] && fwd[2 * (nf - 1) + 1] != rev[1])) {   /* :1335-1337 */
            uint64_t a1 = nf ? fwd[2 * (nf - 1)] : 0, b1 = nr ? rev[0] : 0;
            const uint64_t* from1 = nf ? &a1 : s1; const uint64_t nfrom1 = nf ? 1 : g1->n_src;
            const uint64_t* to1 = nr ? &b1 : k1; const uint64_t nto1 = nr ? 1 : g1->n_snk;
            if (nfrom1 && nto1) np1 = sp_sets(g1, &nx1, ord1, from1, nfrom1, to1, nto1, dp, p1);
            if (np1) {
                uint64_t a2 = nf ? fwd[2 * (nf - 1) + 1] : 0, b2 = nr ? rev[1] : 0;
                const uint64_t* from2 = nf ? &a2 : s2; const uint64_t nfrom2 = nf ? 1 : g2->n_src;
                const uint64_t* to2 = nr ? &b2 : k2; const uint64_t nto2 = nr ? 1 : g2->n_snk;
                if (nfrom2 && nto2) np2 = sp_sets(g2, &nx2, ord2, from2, nfrom2, to2, nto2, dp, p2);
                if (np2) {
                    found = 1;
                    if (nf) { memmove(p1, p1 + 1, (np1 - 1) * 8); --np1; memmove(p2, p2 + 1, (np2 - 1) * 8); --np2; }
                    if (nr) { --np1; --np2; }
                }
            }
        }
        if (!found) {
            /* bisect on the total trim (:1515-1546); reachability == a non-empty shortest path (what both the direct test
             * and the distance oracle of :1477-1497 decide) */
            int64_t lo = 1, hi = (int64_t)(nf + nr);
            while (lo <= hi) {
                const int64_t total = (lo + hi) / 2;
                int success = 0;
                const uint64_t lmin = total > (int64_t)nr ? (uint64_t)(total - (int64_t)nr) : 0;
                const uint64_t lmax = (uint64_t)total < nf ? (uint64_t)total : nf;
                for (uint64_t l = lmin; l <= lmax && !success; ++l) {
                    const uint64_t r = (uint64_t)total - l;
                    /* test_reachability(l, r), :1435-1511 */
                    const int left_all = l == nf, right_all = r == nr;
                    const int allow_equal = left_all || right_all;
                    const uint64_t nl1 = left_all ? g1->n_src : 1, nl2 = left_all ? g2->n_src : 1;
                    const uint64_t nr1 = right_all ? g1->n_snk : 1, nr2 = right_all ? g2->n_snk : 1;
                    for (uint64_t a = 0; a < nl1 && !success; ++a)
                        for (uint64_t b = 0; b < nl2 && !success; ++b)
                            for (uint64_t c = 0; c < nr1 && !success; ++c)
                                for (uint64_t d = 0; d < nr2 && !success; ++d) {
                                    const uint64_t L1 = left_all ? s1[a] : fwd[2 * (nf - 1 - l)], L2 = left_all ? s2[b] : fwd[2 * (nf - 1 - l) + 1];
                                    const uint64_t R1 = right_all ? k1[c] : rev[2 * r], R2 = right_all ? k2[d] : rev[2 * r + 1];
                                    if (!allow_equal && (L1 == R1 || L2 == R2)) continue;
                                    if (sp_sets(g1, &nx1, ord1, &L1, 1, &R1, 1, dp, tmp) && sp_sets(g2, &nx2, ord2, &L2, 1, &R2, 1, dp, tmp)) success = 1;
                                }
                    if (success) { left_trim = l; right_trim = r; }
                }
                if (success) hi = total - 1;
                else lo = total + 1;
            }
            uint64_t a1 = 0, a2 = 0, b1 = 0, b2 = 0;
            const int left_all = left_trim == nf, right_all = right_trim == nr;
            if (!left_all) { a1 = fwd[2 * (nf - left_trim - 1)]; a2 = fwd[2 * (nf - left_trim - 1) + 1]; }
            if (!right_all) { b1 = rev[2 * right_trim]; b2 = rev[2 * right_trim + 1]; }
            np1 = sp_sets(g1, &nx1, ord1, left_all ? s1 : &a1, left_all ? g1->n_src : 1, right_all ? k1 : &b1, right_all ? g1->n_snk : 1, dp, p1);
            np2 = sp_sets(g2, &nx2, ord2, left_all ? s2 : &a2, left_all ? g2->n_src : 1, right_all ? k2 : &b2, right_all ? g2->n_snk : 1, dp, p2);
            if (!left_all) { if (np1) { memmove(p1, p1 + 1, (np1 - 1) * 8); --np1; } if (np2) { memmove(p2, p2 + 1, (np2 - 1) * 8); --np2; } }
            if (!right_all) { if (np1) --np1; if (np2) --np2; }
        }
        uint64_t np = 0;
        for (uint64_t i = 0; i + left_trim < nf; ++i) { pairs_out[2 * np] = fwd[2 * i]; pairs_out[2 * np + 1] = fwd[2 * i + 1]; ++np; }
        for (uint64_t i = 0; i < np1; ++i) { pairs_out[2 * np] = p1[i]; pairs_out[2 * np + 1] = CL_GAP; ++np; }
        for (uint64_t i = 0; i < np2; ++i) { pairs_out[2 * np] = CL_GAP; pairs_out[2 * np + 1] = p2[i]; ++np; }
        for (uint64_t i = right_trim; i < nr; ++i) { pairs_out[2 * np] = rev[2 * i]; pairs_out[2 * np + 1] = rev[2 * i + 1]; ++np; }
        *n_pairs_out = np;
    }
done:
    free(ord1); free(ord2); free(back); free(stack); free(fwd); free(rev); free(dp); free(p1); free(p2); free(tmp);
    free(s1); free(s2); free(k1); free(k2);
    free_adj(&nx1); free_adj(&nx2);
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
        case CL_ROUTE_GREEDY_PARTIAL:
            rc = clo_greedy_partial_alignment(&g1, &g2, pairs, &np);   /* do_alignment does not ask for a score */
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
