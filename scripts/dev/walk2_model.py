# model of the walk2 coverage: main window + helpers; checks every (record, query) with record finalised in an earlier step is covered
import random
def run(count, groups, WIN, S):
    # groups: list of group end indices (exclusive), increasing, last == count
    gend = [0]*count
    a = 0
    for e in groups:
        for t in range(a, e): gend[t] = e
        a = e
    BW = WIN // S
    NS = WIN  # slots
    qi = list(range(NS))          # query index per slot
    cover = {}                     # (rec, query) -> count by main
    ci = 0
    steps = 0
    entered_at = {q: 0 for q in range(min(NS, count))}
    while ci < count:
        wb = ci - ci % S
        ge = min(gend[ci], wb + WIN)
        # finalise [ci, ge): all must be in window
        for t in range(ci, ge):
            assert qi[t % NS] == t, (t, qi[t % NS], ci, ge)
        nwb = ge - ge % S
        for s in range(NS):
            if qi[s] < nwb:
                qi[s] += WIN
        # accumulate
        for s in range(NS):
            q = qi[s]
            if q < count and q >= ge:
                for l in range(ci, ge):
                    cover[(l, q)] = cover.get((l, q), 0) + 1
        ci = ge
        steps += 1
    # helper coverage: query in sub-block m >= BW gets records [0, (m-BW+1)*S)
    step_of = {}
    ci = 0; k = 0
    while ci < count:
        wb = ci - ci % S
        ge = min(gend[ci], wb + WIN)
        for t in range(ci, ge): step_of[t] = k
        ci = ge; k += 1
    for q in range(count):
        m = q // S
        bound = (m - BW + 1) * S if m >= BW else 0
        for l in range(count):
            need = step_of[l] < step_of[q]
            got = (l < bound) or ((l, q) in cover)
            if need and not got:
                return "MISSING rec %d query %d (m %d bound %d)" % (l, q, m, bound)
            if l < bound and not (step_of[l] < step_of[q]):
                return "helper record %d not final before query %d" % (l, q)
    return steps
random.seed(1)
for trial in range(300):
    count = random.choice([1024, 1000, 37, 256, 257, 513])
    WIN, S = random.choice([(128, 16), (256, 32), (256, 16), (128, 32), (256, 64)])
    groups = []
    a = 0
    big = random.random() < 0.2
    while a < count:
        a = min(count, a + (random.randint(1, 400) if big else random.randint(1, 90)))
        groups.append(a)
    r = run(count, groups, WIN, S)
    assert isinstance(r, int), (r, count, WIN, S, groups[:10])
print("ok")
