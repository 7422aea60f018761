"""Lane-level model of popoa_lane_kernel (popoa_lane.hip): the register / DPP systolic sweep of popoa_linear_kernel extended to NEAR-CHAIN graph pairs — every node
has one or two predecessors (a source's boundary index counted as one); row predecessors lie at most DR ranks back (they arrive on a CONVEYOR of DPP moves,
one lane per step), column predecessors at most DC columns back (per-lane register history) or in a SAVED column (LDS, [slot][row]); boundary row and column in
closed form from the shortest source distance of a node.  The model moves data exactly as the kernel does (what lane l holds at step t, what is shifted in, what a
strip hands to the next) and is checked against the plain pull-form DP of SURVEY.md Appendix A on random near-chain pairs:  python scripts/dev/nearchain_model.py [cases]"""
import sys

import numpy as np

NEG = -(2 ** 30)   # cell_t::mininf = INT32_MIN / 2
BND = 0xFF
LANES = 64


def pull_form(gr, gc, P):
    """reference DP (Appendix A) on rank-space graphs: g = dict(n, preds[1-based lists], src[bool], lab).  Rows = graph R, columns = graph C.
    returns M, V (row-consuming gap family), H planes as (nR+1) x (nC+1) arrays; index 0 = boundary"""
    nR, nC, K = gr["n"], gc["n"], len(P["oe"])
    M = np.full((nR + 1, nC + 1), NEG, np.int64)
    V = np.full((K, nR + 1, nC + 1), NEG, np.int64)
    H = np.full((K, nR + 1, nC + 1), NEG, np.int64)
    Mf = lambda a, b: 0 if (a == 0 and b == 0) else M[a, b]
    for a in range(1, nR + 1):
        for k in range(K):
            v = -P["oe"][k] if gr["src"][a] else NEG
            for p in gr["preds"][a]:
                v = max(v, V[k, p, 0] - P["ext"][k])
            V[k, a, 0] = v
        M[a, 0] = max(V[k, a, 0] for k in range(K))
    for b in range(1, nC + 1):
        for k in range(K):
            v = -P["oe"][k] if gc["src"][b] else NEG
            for q in gc["preds"][b]:
                v = max(v, H[k, 0, q] - P["ext"][k])
            H[k, 0, b] = v
        M[0, b] = max(H[k, 0, b] for k in range(K))
    for a in range(1, nR + 1):
        P1 = gr["preds"][a] + ([0] if gr["src"][a] else [])
        for b in range(1, nC + 1):
            P2 = gc["preds"][b] + ([0] if gc["src"][b] else [])
            s = P["match"] if gr["lab"][a] == gc["lab"][b] else -P["mismatch"]
            m = NEG
            for p in P1:
                for q in P2:
                    m = max(m, Mf(p, q) + s)
            for k in range(K):
                v = NEG
                for p in P1:
                    v = max(v, M[p, b] - P["oe"][k])
                    if p != 0:
                        v = max(v, V[k, p, b] - P["ext"][k])
                h = NEG
                for q in P2:
                    h = max(h, M[a, q] - P["oe"][k])
                    if q != 0:
                        h = max(h, H[k, a, q] - P["ext"][k])
                V[k, a, b], H[k, a, b] = v, h
                m = max(m, v, h)
            M[a, b] = m
    return M, V, H


def source_distance(g):
    """nodes on the shortest walk from a source to the node, the node included (what the boundary cells are a closed form of)"""
    d = [0] * (g["n"] + 1)
    for a in range(1, g["n"] + 1):
        best = 1 if g["src"][a] else 1 << 60
        for p in g["preds"][a]:
            best = min(best, d[p] + 1)
        d[a] = best
    return d


def boundary_gap(P, k, length):
    return -P["oe"][k] - (length - 1) * P["ext"][k]


def boundary_m(P, length):
    return max(boundary_gap(P, k, length) for k in range(len(P["oe"])))


def pack(gr, gc, DR, DC, max_slots=8):
    """host side: row records (two predecessor codes: distance 1..DR or BND), column records (distance 1..DC, 0x80 | slot, or BND; keep flag + slot), saved columns.
    None when the pair is not a near-chain pair in this sense"""
    rows, cols, saved = [None], [None], []
    for a in range(1, gr["n"] + 1):
        codes = [a - p for p in gr["preds"][a]] + ([BND] if gr["src"][a] else [])
        if not 1 <= len(codes) <= 2 or any(c != BND and c > DR for c in codes):
            return None
        rows.append((codes[0], codes[-1]))
    far = set()
    for b in range(1, gc["n"] + 1):
        codes = [b - q for q in gc["preds"][b]] + ([BND] if gc["src"][b] else [])
        if not 1 <= len(codes) <= 2:
            return None
        for q in gc["preds"][b]:
            if b - q > DC:
                far.add(q)
    saved = sorted(far)
    if len(saved) > max_slots:
        return None
    for b in range(1, gc["n"] + 1):
        codes = []
        for q in gc["preds"][b]:
            codes.append(b - q if b - q <= DC else 0x80 | saved.index(q))
        if gc["src"][b]:
            codes.append(BND)
        cols.append((codes[0], codes[-1], saved.index(b) if b in far else None))
    return dict(rows=rows, cols=cols, saved=saved)


def lane_model(gr, gc, P, DR=2, DC=2, W=1, chunk=32, lag=3):
    """the sweep as the kernel runs it: strips of 64 rows, strip s on wave s % W, chunks of `chunk` steps, strip s + 1 `lag` chunks behind strip s; returns planes"""
    pk = pack(gr, gc, DR, DC)
    assert pk is not None
    nR, nC, K = gr["n"], gc["n"], len(P["oe"])
    dR, dC = source_distance(gr), source_distance(gc)
    M = np.full((nR + 1, nC + 1), NEG, np.int64)
    V = np.full((K, nR + 1, nC + 1), NEG, np.int64)
    H = np.full((K, nR + 1, nC + 1), NEG, np.int64)
    # boundary cells in closed form (the prologue's cooperative stores)
    for a in range(1, nR + 1):
        for k in range(K):
            V[k, a, 0] = boundary_gap(P, k, dR[a])
        M[a, 0] = boundary_m(P, dR[a])
    for b in range(1, nC + 1):
        for k in range(K):
            H[k, 0, b] = boundary_gap(P, k, dC[b])
        M[0, b] = boundary_m(P, dC[b])
    S = (nR + LANES - 1) // LANES
    Cn = (nC + LANES - 1 + chunk - 1) // chunk
    steps = Cn * chunk
    n_slots = len(pk["saved"])
    lds_saved = {}           # (slot, row) -> (M, [H_k]); row 0 = the boundary row's Mf at that column
    handoff = {}             # (strip, dd, column) -> (M, [V_k]) of the strip's row 64 - dd, written by its last DR lanes
    # every strip's register state lives across its chunks
    state = [None] * S
    # global macro-step order: strip s runs chunk c at macro-step c + lag * s (waves run side by side; a barrier per macro-step).  Modelled sequentially in an
    # order that respects exactly those barriers: macro-step by macro-step, strips in any order inside one.
    for macro in range(Cn + lag * (S - 1)):
        for s in range(S):
            c = macro - lag * s
            if c < 0 or c >= Cn:
                continue
            if c == 0:
                st = dict(lastM=[NEG] * LANES, lastV=[[NEG] * K for _ in range(LANES)],
                          conv=[[(NEG, [NEG] * K) for _ in range(DR + 1)] for _ in range(LANES)],          # conv[l][d], d = 1..DR
                          convMh=[[[NEG] * (DC + 1) for _ in range(DR + 1)] for _ in range(LANES)],       # [l][d][e]
                          Mh=[[NEG] * (DC + 1) for _ in range(LANES)], Hh=[[[NEG] * (DC + 1) for _ in range(K)] for _ in range(LANES)],
                          bMh=[[NEG] * (DC + 1) for _ in range(LANES)], crec=[None] * LANES, cbm=[NEG] * LANES)
                state[s] = st
            st = state[s]
            t0 = c * chunk
            # lanes j < chunk load this chunk's columns (records, boundary-row Mf) and, for s > 0, the hand-off rows: they rotate down one lane per step
            feed_rec = [None] * LANES
            feed_bm = [NEG] * LANES
            feed_conv = [[(NEG, [NEG] * K) for _ in range(DR + 1)] for _ in range(LANES)]
            for j in range(chunk):
                col = t0 + j + 1
                if col <= nC:
                    feed_rec[j] = (col, pk["cols"][col])
                    feed_bm[j] = boundary_m(P, dC[col])
                    if s > 0:
                        for dd in range(1, DR + 1):
                            feed_conv[j][dd] = handoff.get((s - 1, dd, col), (NEG, [NEG] * K))
            for jj in range(chunk):
                t = t0 + jj
                # --- the DPP moves of a step: every lane takes what its upper neighbour holds, lane 0 the head of the feeds; the feeds rotate down ---
                new_conv = [[None] * (DR + 1) for _ in range(LANES)]
                for l in range(LANES):
                    for d in range(1, DR + 1):
                        if l == 0:
                            new_conv[l][d] = feed_conv[0][d]
                        elif d == 1:
                            new_conv[l][d] = (st["lastM"][l - 1], list(st["lastV"][l - 1]))
                        else:
                            new_conv[l][d] = st["conv"][l - 1][d - 1]
                new_crec = [feed_rec[0]] + st["crec"][:-1]
                new_cbm = [feed_bm[0]] + st["cbm"][:-1]
                feed_rec = feed_rec[1:] + [None]
                feed_bm = feed_bm[1:] + [NEG]
                feed_conv = feed_conv[1:] + [[(NEG, [NEG] * K) for _ in range(DR + 1)]]
                # histories move one step: what was "now" becomes "one step ago" (own row, conveyor M, boundary-row Mf)
                for l in range(LANES):
                    for e in range(DC, 1, -1):
                        st["bMh"][l][e] = st["bMh"][l][e - 1]
                        for d in range(1, DR + 1):
                            st["convMh"][l][d][e] = st["convMh"][l][d][e - 1]
                    # e = 1: the values of the previous step
                    st["bMh"][l][1] = st["cbm"][l]
                    for d in range(1, DR + 1):
                        st["convMh"][l][d][1] = st["conv"][l][d][0]
                    # (the lane's own history moves when the lane has computed its cell: below)
                st["conv"], st["crec"], st["cbm"] = new_conv, new_crec, new_cbm
                for l in range(LANES):
                    a = s * LANES + l + 1
                    b = t - l + 1
                    if not (1 <= b <= nC and a <= nR):
                        continue
                    rec = st["crec"][l]
                    assert rec is not None and rec[0] == b, (s, l, t, rec, b)
                    e0, e1, keep = rec[1]
                    r0, r1 = pk["rows"][a]
                    s_ab = P["match"] if gr["lab"][a] == gc["lab"][b] else -P["mismatch"]
                    own_bnd = boundary_m(P, dR[a])
                    v = [NEG] * K
                    h = [NEG] * K
                    m = NEG
                    for rc in (r0, r1):
                        if rc == BND:
                            mu, vu = st["cbm"][l], None
                        else:
                            mu, vu = st["conv"][l][rc]
                        for k in range(K):
                            v[k] = max(v[k], mu - P["oe"][k])
                            if vu is not None:
                                v[k] = max(v[k], vu[k] - P["ext"][k])
                        for ec in (e0, e1):
                            if rc == BND and ec == BND:
                                d_ = 0
                            elif rc == BND:
                                d_ = lds_saved[(ec & 0x7F, 0)][0] if ec & 0x80 else st["bMh"][l][ec]
                            elif ec == BND:
                                d_ = boundary_m(P, dR[a - rc])
                            elif ec & 0x80:
                                d_ = lds_saved[(ec & 0x7F, a - rc)][0]
                            else:
                                d_ = st["convMh"][l][rc][ec]
                            m = max(m, d_ + s_ab)
                    for ec in (e0, e1):
                        if ec == BND:
                            ml, hl = own_bnd, None
                        elif ec & 0x80:
                            ml, hl = lds_saved[(ec & 0x7F, a)]
                        else:
                            ml, hl = st["Mh"][l][ec], [st["Hh"][l][k][ec] for k in range(K)]
                        for k in range(K):
                            h[k] = max(h[k], ml - P["oe"][k])
                            if hl is not None:
                                h[k] = max(h[k], hl[k] - P["ext"][k])
                    for k in range(K):
                        m = max(m, v[k], h[k])
                    M[a, b] = m
                    for k in range(K):
                        V[k, a, b], H[k, a, b] = v[k], h[k]
                    # the new cell becomes the lane's "previous column"; the lane below takes M / V on the next step
                    st["lastM"][l] = m
                    st["lastV"][l] = list(v)
                    for e in range(DC, 1, -1):   # own history: after the reads of this step
                        st["Mh"][l][e] = st["Mh"][l][e - 1]
                        for k in range(K):
                            st["Hh"][l][k][e] = st["Hh"][l][k][e - 1]
                    st["Mh"][l][1] = m
                    for k in range(K):
                        st["Hh"][l][k][1] = h[k]
                    if keep is not None:
                        lds_saved[(keep, a)] = (m, list(h))
                        if a == 1:
                            lds_saved[(keep, 0)] = (st["cbm"][l], None)
                    if s + 1 < S and l >= LANES - DR:
                        handoff[(s, LANES - l, b)] = (m, list(v))
    return M, V, H


def random_near_chain(rng, n, DR_or_DC, n_far=0, p_bubble=0.15, alt_src=True):
    """a rank-space graph: chain with short bubbles / skips whose predecessors lie within the given distance, plus n_far long-range forks (columns only)"""
    preds = [None, []]
    src = [False, True]
    for a in range(2, n + 1):
        x = rng.random()
        if x < p_bubble and a > 2:
            d2 = int(rng.integers(2, min(DR_or_DC, a - 1) + 1)) if min(DR_or_DC, a - 1) >= 2 else 1
            ps = sorted({a - 1, a - d2})
            if rng.random() < 0.5 and len(ps) == 2:
                ps = [ps[1], ps[0]]
            if rng.random() < 0.3:
                ps = [a - d2]          # the alternate branch alone: the node skips its rank predecessor (a bubble's second arm)
        else:
            ps = [a - 1]
        preds.append(ps)
        src.append(bool(alt_src and a <= 4 and rng.random() < 0.3 and len(ps) == 1))
    for _ in range(n_far):
        a = int(rng.integers(max(3, DR_or_DC + 2), n + 1))
        q = int(rng.integers(1, a - DR_or_DC)) if a - DR_or_DC > 1 else None
        if q and len(preds[a]) == 1 and not src[a]:
            preds[a] = [preds[a][0], q] if rng.random() < 0.5 else [q]
    # every node must be reachable and lie on a source->sink walk is not needed for the DP identity; in-degree >= 1 or source is
    lab = [0] + [int(x) for x in rng.integers(1, 5, n)]
    return dict(n=n, preds=preds, src=src, lab=lab)


def main():
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    rng = np.random.default_rng(5)
    params = [dict(match=20, mismatch=80, oe=[90, 805, 2501], ext=[30, 5, 1]), dict(match=1, mismatch=1, oe=[4, 4, 4], ext=[3, 2, 1])]
    bad = 0
    for case in range(cases):
        K = int(rng.integers(1, 4))
        P0 = params[case % 2]
        P = dict(match=P0["match"], mismatch=P0["mismatch"], oe=P0["oe"][:K], ext=P0["ext"][:K])
        DR, DC = int(rng.integers(1, 5)), int(rng.integers(1, 5))
        nR = int(rng.integers(1, 150))
        nC = int(rng.integers(1, 260))
        gr = random_near_chain(rng, nR, DR)
        gc = random_near_chain(rng, nC, DC, n_far=int(rng.integers(0, 4)))
        if pack(gr, gc, DR, DC) is None:
            continue
        want = pull_form(gr, gc, P)
        got = lane_model(gr, gc, P, DR, DC)
        ok = all(np.array_equal(w[..., 1:, 1:], g[..., 1:, 1:]) for w, g in zip(want, got)) and np.array_equal(want[0][:, 0], got[0][:, 0]) and \
            np.array_equal(want[0][0, 1:], got[0][0, 1:]) and np.array_equal(want[1][:, 1:, 0], got[1][:, 1:, 0]) and np.array_equal(want[2][:, 0, 1:], got[2][:, 0, 1:])
        bad += not ok
        print("case %3d: %3d x %3d, NumPW %d, DR %d DC %d, %d saved columns: %s" % (case, nR, nC, K, DR, DC, len(pack(gr, gc, DR, DC)["saved"]), "ok" if ok else "DIFFERENT"), flush=True)
    print("%d different" % bad)
    return bad


if __name__ == "__main__":
    sys.exit(1 if main() else 0)
