"""TEST INFRASTRUCTURE ONLY (oracle/): a pure-Python restatement of the reference's pruned graph-graph wavefront alignment, the route "w" of
Stitcher::do_alignment (include/centrolign/stitcher.hpp:326-339) — pwfa_po_poa, include/centrolign/alignment.hpp:2299-2338, over pwfa_po_poa_internal :1959-2034,
wfa_iteration :1712-1875 (the forward half), wfa_traceback :1892-1923, to_wfa_params :1613-1655, minmax_distance (minmax_distance.hpp:16-72) and target_reachability
(target_reachability.hpp:16-32).  Only tests/ may import it.  Small cases only (Python loops).  Pinned against the compiled reference's outputs for this route in
tests/golden/host_routes.npz (tests/test_wfa_oracle.py); the product's C++ (centrolign_amd/csrc/wfa_host.hpp) is compared with it on fresh random pairs.

What the algorithm is: a shortest-path search over states (node of graph 1, node of graph 2, component) — component 0 = match state, +i / -i = inside a gap of piece
i that consumes graph 1 / graph 2 — with integer edge costs (match 0), served from FIFO buckets by score, so that among equal scores the first state ENQUEUED wins:
the enqueue order below is the reference's.  Pruning: a state is dropped when its nodes cannot reach a sink, or when it lags more than prune_limit behind the furthest
state visited so far (distance = edges from the sources)."""
from collections import deque
from math import gcd

GAP = (1 << 64) - 1


class Side:
    def __init__(self, side, k):
        lo, hi = int(side.node_off[k]), int(side.node_off[k + 1])
        self.n = hi - lo
        self.label = [int(x) for x in side.label[lo:hi]]
        self.next = [[int(x) for x in side.next_idx[int(side.next_off[v]):int(side.next_off[v + 1])]] for v in range(lo, hi)]
        self.prev = [[int(x) for x in side.prev_idx[int(side.prev_off[v]):int(side.prev_off[v + 1])]] for v in range(lo, hi)]
        self.sources = [int(x) for x in side.src_idx[int(side.src_off[k]):int(side.src_off[k + 1])]]
        self.sinks = [int(x) for x in side.snk_idx[int(side.snk_off[k]):int(side.snk_off[k + 1])]]
        # a topological order (any will do for the two distance tables)
        indeg = [len(p) for p in self.prev]
        stack = [v for v in range(self.n) if indeg[v] == 0]
        self.order = []
        while stack:
            v = stack.pop()
            self.order.append(v)
            for w in self.next[v]:
                indeg[w] -= 1
                if indeg[w] == 0:
                    stack.append(w)

    def after(self, v):      # the dummy start (id n) stands in front of the sources
        return self.sources if v == self.n else self.next[v]

    def source_distances(self):          # minmax_distance.hpp:16-72: (fewest, most) edges from a source; unreached: (inf, -1)
        lo, hi = [None] * self.n, [-1] * self.n
        for v in self.sources:
            lo[v], hi[v] = 0, 0
        for v in self.order:
            if lo[v] is None:
                continue
            for w in self.next[v]:
                lo[w] = lo[v] + 1 if lo[w] is None else min(lo[w], lo[v] + 1)
                hi[w] = max(hi[w], hi[v] + 1)
        return lo, hi

    def reaches_sink(self):              # target_reachability.hpp:16-32
        r = [False] * self.n
        for v in self.sinks:
            r[v] = True
        for v in reversed(self.order):
            r[v] = r[v] or any(r[w] for w in self.next[v])
        return r


def wfa_costs(match, mismatch, gap_open, gap_extend, npw):
    """to_wfa_params (:1613-1655): maximising scores -> minimising costs with match = 0, reduced by the common factor"""
    mm = 2 * (match + mismatch)
    go = [2 * gap_open[i] for i in range(npw)]
    ge = [2 * gap_extend[i] + match for i in range(npw)]
    f = mm
    for x in go + ge:
        f = gcd(f, x)
    return mm // f, [x // f for x in go], [x // f for x in ge]


def pwfa_po_poa(g1, g2, costs, prune_limit):
    mm, go, ge = costs
    npw = len(go)
    lo1, hi1 = g1.source_distances()
    lo2, hi2 = g2.source_distances()
    ok1, ok2 = g1.reaches_sink(), g2.reaches_sink()
    sink1, sink2 = set(g1.sinks), set(g2.sinks)
    furthest = [-(1 << 62)]
    back = {}
    buckets = [deque()]          # buckets[i]: states at score floor + i
    floor = 0
    buckets[0].append(((GAP, GAP, 0), (g1.n, g2.n, 0)))

    def put(frm, to, cost):
        while len(buckets) <= cost:
            buckets.append(deque())
        buckets[cost].append((frm, to))

    def lagging(a, b):
        if (a < g1.n and not ok1[a]) or (b < g2.n and not ok2[b]):
            return True
        d1 = hi1[a] if a != g1.n else -1
        d2 = hi2[b] if b != g2.n else -1
        return d1 + d2 < furthest[0] - prune_limit

    while True:
        while not buckets[0]:
            buckets.pop(0)
            floor += 1
        frm, here = buckets[0].popleft()
        a, b, comp = here
        if lagging(a, b) or here in back:
            continue
        if (a == g1.n or ok1[a]) and (b == g2.n or ok2[b]):      # the furthest position by FEWEST edges
            d1 = lo1[a] if a != g1.n else -1
            d2 = lo2[b] if b != g2.n else -1
            furthest[0] = max(furthest[0], d1 + d2)
        back[here] = frm
        if (not sink1 or a in sink1) and (not sink2 or b in sink2) and comp == 0:
            break
        n1s, n2s = g1.after(a), g2.after(b)
        if comp == 0:
            if len(n1s) == 1 and len(n2s) == 1 and a not in sink1 and b not in sink2 and g1.label[n1s[0]] == g2.label[n2s[0]]:
                put(here, (n1s[0], n2s[0], 0), 0)                 # the only way on is a match: nothing else is queued
                continue
            for x in n1s:
                for y in n2s:
                    put(here, (x, y, 0), 0 if g1.label[x] == g2.label[y] else mm)
                for i in range(npw):
                    put(here, (x, b, i + 1), go[i] + ge[i])
            for y in n2s:
                for i in range(npw):
                    put(here, (a, y, -i - 1), go[i] + ge[i])
        else:
            put(here, (a, b, 0), 0)
            if comp > 0:
                for x in n1s:
                    put(here, (x, b, comp), ge[comp - 1])
            else:
                for y in n2s:
                    put(here, (a, y, comp), ge[-comp - 1])
    # wfa_traceback (:1892-1923)
    pairs = []
    a, b, comp = here
    while a != g1.n or b != g2.n:
        pa, pb, pc = back[(a, b, comp)]
        if pa != a and pb != b:
            pairs.append((a, b))
        elif pa != a:
            pairs.append((a, GAP))
        elif pb != b:
            pairs.append((GAP, b))
        a, b, comp = pa, pb, pc
    pairs.reverse()
    return pairs


# ---- the two-sided search around one long deletion: deletion_wfa_po_poa, alignment.hpp:2036-2282 (routes "ad1" / "ad2", stitcher.hpp:284-325) -------------------------
# `sh` is the short graph (the first coordinate of the pairs returned), `lg` the long one.  A front from the sources (forward expansions, as above but unpruned and
# never greedy) and a front from the sink pairs (the mirror-image expansions of wfa_iteration's reverse half, :1800-1875) advance in turn, the one with the lower score
# floor first.  Each records where it is in match state: short node -> [(long node, score)].  As soon as a recorded position of one front can be joined to one of the
# other by a deletion in the long graph (or they coincide), the search is given `scope` more score units — no single step costs more — and then stops.  The junction
# with the lowest total (both scores + the cheapest gap piece for the deleted stretch) is taken; among equal totals the first in the iteration order of the forward
# front's std::unordered_map wins, which oracle/std_unordered_order.py reproduces.

def _hops(lg, frm):
    """edges on a shortest path from `frm` to every node of lg, None where there is none (SuperbubbleDistanceOracle::min_distance as its test defines it)"""
    d = [None] * lg.n
    d[frm] = 0
    for v in lg.order:
        if d[v] is not None:
            for w in lg.next[v]:
                if d[w] is None or d[v] + 1 < d[w]:
                    d[w] = d[v] + 1
    return d


def _shortest_path(lg, frm, to):
    """shortest_path.hpp:32-100: the nodes of a shortest path, walking back from `to` along the first predecessor (in list order) that is one edge nearer"""
    d = _hops(lg, frm)
    if d[to] is None:
        return []
    path = [to]
    while d[path[-1]] != 0:
        for p in lg.prev[path[-1]]:
            if d[p] is not None and d[p] + 1 == d[path[-1]]:
                path.append(p)
                break
        else:
            break
    path.reverse()
    return path


def deletion_wfa_po_poa(sh, lg, costs):
    from oracle.std_unordered_order import UnorderedKeys
    mm, go, ge = costs
    npw = len(go)
    scope = max([mm] + [go[i] + ge[i] for i in range(npw)])
    src_sh, src_lg = set(sh.sources), set(lg.sources)
    hops_memo = {}

    def hops(a, b):
        if a not in hops_memo:
            hops_memo[a] = _hops(lg, a)
        return hops_memo[a][b]

    class Front:
        def __init__(self):
            self.buckets, self.floor, self.back = [deque()], 0, {}
            self.keys, self.landed = UnorderedKeys(), {}

        def put(self, frm, to, cost):
            while len(self.buckets) <= cost:
                self.buckets.append(deque())
            self.buckets[cost].append((frm, to))

        def pop(self):
            while not self.buckets[0]:
                self.buckets.pop(0)
                self.floor += 1
            return self.buckets[0].popleft()

    fwd, rev = Front(), Front()
    fwd.put((GAP, GAP, 0), (sh.n, lg.n, 0), 0)
    for a in sh.sinks:
        for b in lg.sinks:
            rev.put((GAP, GAP, 0), (a, b, 0), 0)
    until = [None]

    def joinable(before, after):
        return before == after or (before != lg.n and after != lg.n and hops(before, after) is not None)

    def settle(mine, other, here, mine_is_before):
        a, b, comp = here
        if comp == 0:
            mine.keys.insert(a)
            mine.landed.setdefault(a, []).append((b, mine.floor))
        if until[0] is None and a in other.landed:
            for b2, _ in other.landed[a]:
                if joinable(b, b2) if mine_is_before else joinable(b2, b):
                    until[0] = mine.floor + scope

    def finished():
        return until[0] is not None and fwd.floor >= until[0] and rev.floor >= until[0]

    while True:
        if fwd.floor <= rev.floor:
            frm, here = fwd.pop()
            if here in fwd.back:
                continue
            settle(fwd, rev, here, True)
            fwd.back[here] = frm
            if finished():
                break
            a, b, comp = here
            n1s, n2s = sh.after(a), lg.after(b)
            if comp == 0:
                for x in n1s:
                    for y in n2s:
                        fwd.put(here, (x, y, 0), 0 if sh.label[x] == lg.label[y] else mm)
                    for i in range(npw):
                        fwd.put(here, (x, b, i + 1), go[i] + ge[i])
                for y in n2s:
                    for i in range(npw):
                        fwd.put(here, (a, y, -i - 1), go[i] + ge[i])
            else:
                fwd.put(here, (a, b, 0), 0)
                if comp > 0:
                    for x in n1s:
                        fwd.put(here, (x, b, comp), ge[comp - 1])
                else:
                    for y in n2s:
                        fwd.put(here, (a, y, comp), ge[-comp - 1])
        else:
            frm, here = rev.pop()
            if here in rev.back:
                continue
            settle(rev, fwd, here, False)
            rev.back[here] = frm
            if finished():
                break
            a, b, comp = here
            p1s = (sh.prev[a] + ([sh.n] if a in src_sh else [])) if a < sh.n else []
            p2s = (lg.prev[b] + ([lg.n] if b in src_lg else [])) if b < lg.n else []
            if comp == 0:
                if a < sh.n and b < lg.n:
                    cost = 0 if sh.label[a] == lg.label[b] else mm
                    for x in p1s:
                        for y in p2s:
                            rev.put(here, (x, y, 0), cost)
                for i in range(npw):
                    rev.put(here, (a, b, i + 1), 0)
                    rev.put(here, (a, b, -i - 1), 0)
            elif comp > 0:
                for x in p1s:
                    rev.put(here, (x, b, comp), ge[comp - 1])
                    rev.put(here, (x, b, 0), go[comp - 1] + ge[comp - 1])
            else:
                for y in p2s:
                    rev.put(here, (a, y, comp), ge[-comp - 1])
                    rev.put(here, (a, y, 0), go[-comp - 1] + ge[-comp - 1])
    best = None
    for a in fwd.keys:
        if a not in rev.landed:
            continue
        for bf, sf in fwd.landed[a]:
            if bf == lg.n:
                continue
            for br, sr in rev.landed[a]:
                if br == lg.n:
                    continue
                h = hops(bf, br)
                if h is None:
                    continue
                total = min(go[i] + ge[i] * h for i in range(npw)) + sf + sr
                if best is None or total < best[0]:
                    best = (total, a, bf, br)
    _, a0, bf, br = best
    pairs = []
    a, b, comp = a0, bf, 0
    while a != sh.n or b != lg.n:                         # the left half, back to the start (wfa_traceback)
        pa, pb, pc = fwd.back[(a, b, comp)]
        pairs.append((a, b) if pa != a and pb != b else (a, GAP) if pa != a else (GAP, b) if pb != b else None)
        a, b, comp = pa, pb, pc
    pairs = [p for p in reversed(pairs) if p is not None]
    for v in _shortest_path(lg, bf, br)[1:]:              # the deleted stretch of the long graph
        pairs.append((GAP, v))
    a, b, comp = a0, br, 0
    nxt = rev.back[(a, b, comp)]                          # the right half, on to the sinks (wfa_traceback_rev, :1925-1957)
    while nxt[0] != GAP and nxt[1] != GAP:
        if nxt[0] != a and nxt[1] != b:
            pairs.append((nxt[0], nxt[1]))
        elif nxt[0] != a:
            pairs.append((nxt[0], GAP))
        elif nxt[1] != b:
            pairs.append((GAP, nxt[1]))
        a, b, comp = nxt
        nxt = rev.back[(a, b, comp)]
    return pairs
