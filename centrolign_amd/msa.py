"""Driver of a progressive MSA over the C ABI — plumbing for the tests, the bench and scripts/native_*.py, standing where the
reference's Tree / Execution stand (src/execution.cpp:9-133): it walks a caller-supplied binary guide tree (nested 2-tuples of
sequence names) and calls, per leaf, cl_leaf_graph + cl_leaf_intrinsic_scale and, per internal node, cl_merge.  Node ids of the
reference's Tree follow the Newick text (src/tree.cpp:75-132), so the leaves are calibrated in order of appearance and the first child
of a node is graph 1 of its merge."""
from . import capi


def balanced_tree(names):
    if len(names) == 1:
        return names[0]
    h = len(names) // 2
    return (balanced_tree(names[:h]), balanced_tree(names[h:]))


def newick(tree):
    return tree if isinstance(tree, str) else "(" + newick(tree[0]) + "," + newick(tree[1]) + ")"


def leaves_of(tree):
    return [tree] if isinstance(tree, str) else leaves_of(tree[0]) + leaves_of(tree[1])


def tree_of_plan(newick_text, names):
    """the guide tree as the nested tuple the drivers here take, in the reference's orientation: Tree(newick) with Execution's pruning,
    binarisation and child order (capi.msa_plan): every merge is (graph 1, graph 2) as Execution::next hands them out"""
    leaves, merges = capi.msa_plan(newick_text, list(names))
    slots = [names[i] for i in leaves]
    for a, b in merges:
        slots.append((slots[a], slots[b]))
    return slots[-1]


def subproblem_file_name(prefix, names):
    """Core::subproblem_file_name (src/core.cpp:378-380): PREFIX_<hash of the sorted leaf names>.gfa"""
    return "%s_%s.gfa" % (prefix, capi.subproblem_hash_hex(list(names)))


def emit_subproblem(prefix, graph, paths):
    """Core::emit_subproblem (src/core.cpp:397-422): the subproblem's GFA under its hashed name, one line in PREFIX_info.txt"""
    import os
    name = subproblem_file_name(prefix, paths)
    info = prefix + "_info.txt"
    header = not os.path.exists(info)
    with open(info, "a") as f:
        if header:
            f.write("filename\tsequences\n")
        f.write("%s\t%s\n" % (name, ",".join(sorted(paths))))
    with open(name, "wb") as f:
        f.write(capi.write_gfa(graph, paths))


def progressive_msa(ctx, sequences, tree, max_num_match_pairs=1250000, max_count=3000, workers=1, make_context=None, verbose=False,
                    keep_merges=False, subproblems_prefix=None, restart=False, devices=None, calibration_contexts=None):
    """sequences: {name: str}.  Returns dict(root BaseGraph, paths [names in path order], alignment of the root merge, scale,
    scales, stats).  workers > 1: independent pieces of the job (the leaf calibrations; sibling merges of the guide tree) run
    side by side on one device, each worker thread with its own cl_context (the library calls release the GIL): one
    anchor chain leaves much of the device idle between its host phases, a second one fills the gaps."""
    import time as _time
    _t0 = _time.perf_counter()
    order = leaves_of(tree)
    if len(order) > 1 and int(workers) > 1:
        from concurrent.futures import ThreadPoolExecutor
        with ThreadPoolExecutor(min(len(order), 8)) as pool:      # (the ABI call releases the GIL; 25 ms per Mbp each)
            leaves = dict(zip(order, pool.map(lambda nm: capi.leaf_graph(sequences[nm]), order)))
    else:
        leaves = {nm: capi.leaf_graph(sequences[nm]) for nm in order}
    # devices = [ordinals]: worker w sits on devices[w % len(devices)] — one process, several GPUs (graphs are host arrays at the C ABI, so
    # nothing travels between the devices but the calls); worker 0 is ctx itself
    if make_context is None and devices:
        _next = iter(range(1, 1 << 30))
        make_context = lambda: capi.Context(int(devices[next(_next) % len(devices)]))   # noqa: E731
    make_context = make_context or (lambda: capi.Context(getattr(ctx, "device", 0)))
    contexts = [ctx] + [make_context() for _ in range(max(1, int(workers)) - 1)]
    stats = dict(match_ms=0.0, align_ms=0.0, fuse_ms=0.0, merges=0)
    last = {}

    def in_parallel(jobs, contexts=contexts):
        """jobs: list of callables taking a context; results in order"""
        if len(contexts) == 1 or len(jobs) <= 1:
            return [job(contexts[0]) for job in jobs]
        import queue
        from concurrent.futures import ThreadPoolExecutor
        free = queue.Queue()
        for c in contexts:
            free.put(c)

        def run(job):
            c = free.get()
            try:
                return job(c)
            finally:
                free.put(c)
        with ThreadPoolExecutor(len(contexts)) as pool:
            return list(pool.map(run, jobs))

    stats["timeline_s"] = [("leaf graphs done", _time.perf_counter() - _t0)]
    try:
        # the calibrations are as many independent jobs as there are leaves and every merge waits for the mean of ALL scales; calibration_contexts = n gives
        # them up to n contexts of their own for the duration instead of the merge workers'.  Off by default — measured (10 x 1 Mbp, four merge workers,
        # scripts/dev/msa_calib_ab.py): the ten calibrations are done after 1.25 s on the four workers' contexts, after 1.26-1.38 s on eight, 1.56 s on ten:
        # what they queue up for is not contexts (launches of many queues side by side slow one another)
        calib = contexts
        extra = []
        if len(contexts) > 1 and len(order) > len(contexts) and calibration_contexts:
            want = min(len(order), int(calibration_contexts))
            try:
                extra = [make_context() for _ in range(max(0, want - len(contexts)))]
            except Exception:   # noqa: BLE001 (no more contexts to be had: the merge workers' ones do)
                extra = []
            calib = contexts + extra
        try:
            scales = in_parallel([lambda c, nm=nm: c.leaf_intrinsic_scale(leaves[nm], max_count=max_count, max_num_match_pairs=max_num_match_pairs)
                                  for nm in order], calib)
        finally:
            for c in extra:
                if hasattr(c, "close"):
                    c.close()
        scale = sum(scales) / len(scales)                    # ScoreFunction::score_scale (src/core.cpp:169-184)
        stats["timeline_s"].append(("calibrations done", _time.perf_counter() - _t0))

        # the guide tree in waves: every merge whose two children are there runs in the same wave
        done = {nm: (leaves[nm], [nm]) for nm in order}       # subtree (as its newick text) -> (graph, path names)
        pending = []

        def collect(t):
            if not isinstance(t, str):
                collect(t[0]); collect(t[1]); pending.append(t)
        collect(tree)
        if restart and subproblems_prefix:
            # Execution::restart (src/execution.cpp:222-277): top-down, the first finished subproblem on every branch is loaded from its
            # file — read_gfa + add_sentinels, i.e. NOT the node numbering the interrupted run had in memory — and everything below it
            # is skipped
            import os

            def load(t):
                if isinstance(t, str):
                    return
                name = subproblem_file_name(subproblems_prefix, leaves_of(t))
                if os.path.exists(name):
                    with open(name, "rb") as f:
                        done[newick(t)] = capi.read_gfa(f.read())
                    stats["restarted"] = stats.get("restarted", 0) + 1
                else:
                    load(t[0]); load(t[1])
            load(tree)

            def below_loaded(t, under=False):
                out = set()
                if not isinstance(t, str):
                    here = under or newick(t) in done
                    if under or newick(t) in done:
                        out.add(newick(t))
                    out |= below_loaded(t[0], here) | below_loaded(t[1], here)
                return out
            skip = below_loaded(tree)
            pending = [t for t in pending if newick(t) not in skip]
        while pending:
            ready = [t for t in pending if newick(t[0]) in done and newick(t[1]) in done]
            pending = [t for t in pending if not (newick(t[0]) in done and newick(t[1]) in done)]

            def job(c, t):
                (g1, p1), (g2, p2) = done[newick(t[0])], done[newick(t[1])]
                return c.merge(g1, g2, score_scale=scale, max_num_match_pairs=max_num_match_pairs, max_count=max_count), g1, g2, p1 + p2
            for t, (r, g1, g2, paths) in zip(ready, in_parallel([lambda c, t=t: job(c, t) for t in ready])):
                for k in ("match_ms", "align_ms", "fuse_ms"):
                    stats[k] += r[k]
                stats["merges"] += 1
                al = r.get("align") or {}
                if keep_merges:   # what the bench replays: the merge's two graphs and the anchor segments that were stitched
                    stats.setdefault("kept", []).append(dict(merge=newick(t), graphs=(g1, g2), align=al, fused=r["fused"], paths=paths))
                stats.setdefault("per_merge", []).append(dict(merge=newick(t), paths1=len(done[newick(t[0])][1]), paths2=len(done[newick(t[1])][1]),
                                                              match_sets=r["n_match_sets"], match_ms=r["match_ms"], align_ms=r["align_ms"],
                                                              fuse_ms=r["fuse_ms"], nodes=len(r["fused"].label),
                                                              **{k: al.get(k, 0) for k in ("chain_ms", "partition_ms", "stitch_ms", "chain_device_ms",
                                                                                          "chain_pair_evals", "chain_match_pairs", "chain_combinations")}))
                if verbose:
                    import sys
                    print("merge %s: %s" % (newick(t), stats["per_merge"][-1]), file=sys.stderr, flush=True)
                done[newick(t)] = (r["fused"], paths)
                if subproblems_prefix:
                    emit_subproblem(subproblems_prefix, r["fused"], paths)
                last["alignment"], last["graphs"] = r["alignment"], (g1, g2)
            stats["timeline_s"].append(("wave of %d merge(s) done" % len(ready), _time.perf_counter() - _t0))
        root, paths = done[newick(tree)]
    finally:
        for c in contexts[1:]:
            if hasattr(c, "close"):
                c.close()
    return dict(root=root, paths=paths, alignment=last.get("alignment"), root_inputs=last.get("graphs"), scale=scale, scales=scales,
                stats=stats, leaves=leaves)


def output_text(result):
    """what the CLI prints: the explicit CIGAR of a pairwise run, the GFA of an MSA (src/main.cpp)"""
    if len(result["paths"]) == 2:
        g1, g2 = result["root_inputs"]
        return capi.explicit_cigar(g1, g2, result["alignment"])
    return capi.write_gfa(result["root"], result["paths"])


# ---------------------------------------------------------------------------------------------------------------------
# The same MSA over several ranks (one process per GPU).  The path shards at two levels with no collective inside the
# work: the leaf calibrations are independent, and so are sibling merges of the guide tree.  What does cross
# ranks is the one real exchange step of a progressive MSA: a merge needs the fused graphs of its two children, so the
# owner of the right child sends its graph to the owner of the left child.  Graphs are host arrays at the C-ABI boundary, so
# the transport is a host-side point-to-point (a gloo group), not a device collective.

def _count_leaves(tree):
    return 1 if isinstance(tree, str) else _count_leaves(tree[0]) + _count_leaves(tree[1])


def subtree_work(tree):
    """relative cost of building a subtree's graph: a merge's chaining DP grows with the number of (chain of graph 1, chain of
    graph 2) combinations, (paths1 + 1)(paths2 + 1); leaves cost nothing"""
    if isinstance(tree, str):
        return 0
    nl, nr = _count_leaves(tree[0]), _count_leaves(tree[1])
    return (nl + 1) * (nr + 1) + subtree_work(tree[0]) + subtree_work(tree[1])


def split_ranks(tree, ranks):
    """ranks of the left / right child of an internal node (len(ranks) >= 2): in proportion to the work under each child, at least
    one each, and a leaf never more than one (a leaf has nothing to share; surplus ranks go to its sibling)"""
    wl, wr = subtree_work(tree[0]), subtree_work(tree[1])
    n = len(ranks)
    if isinstance(tree[0], str):
        k = 1
    elif isinstance(tree[1], str):
        k = n - 1
    else:
        k = max(1, min(n - 1, round(n * wl / (wl + wr))))
    return ranks[:k], ranks[k:]


def _pack_graph(g):
    import numpy as np
    head = np.array([len(getattr(g, k)) for k in capi.GRAPH_KEYS] + [g.src_id, g.snk_id], np.int64)
    body = b"".join(np.ascontiguousarray(getattr(g, k)).tobytes() for k in capi.GRAPH_KEYS)
    return head, np.frombuffer(body, np.uint8)


def _unpack_graph(head, body):
    import numpy as np
    dts = (np.uint8, np.uint64, np.uint32, np.uint64, np.uint32, np.uint64, np.uint32)
    arrays, at = [], 0
    for n, dt in zip(head[:7], dts):
        nbytes = int(n) * np.dtype(dt).itemsize
        arrays.append(np.frombuffer(body[at:at + nbytes].tobytes(), dt).copy())
        at += nbytes
    return capi.BaseGraph(*arrays, int(head[7]), int(head[8]))


def send_graph(g, dst, dist, group=None):
    import torch
    head, body = _pack_graph(g)
    dist.send(torch.from_numpy(head.copy()), dst, group=group)
    dist.send(torch.from_numpy(body.copy()), dst, group=group)


def recv_graph(src, dist, group=None):
    import numpy as np
    import torch
    head = torch.zeros(9, dtype=torch.int64)
    dist.recv(head, src, group=group)
    sizes = (1, 8, 4, 8, 4, 8, 4)
    body = torch.zeros(int(sum(int(n) * s for n, s in zip(head[:7].tolist(), sizes))), dtype=torch.uint8)
    dist.recv(body, src, group=group)
    return _unpack_graph(head.numpy(), body.numpy())


def _in_parallel(contexts, jobs):
    """jobs: callables taking a context, run side by side on the contexts (one thread per context; the ABI calls release the GIL);
    results in order"""
    if len(contexts) == 1 or len(jobs) <= 1:
        return [job(contexts[0]) for job in jobs]
    import queue
    from concurrent.futures import ThreadPoolExecutor
    free = queue.Queue()
    for c in contexts:
        free.put(c)

    def run(job):
        c = free.get()
        try:
            return job(c)
        finally:
            free.put(c)
    with ThreadPoolExecutor(len(contexts)) as pool:
        return list(pool.map(run, jobs))


def progressive_msa_distributed(ctx, sequences, tree, dist, rank, world, group=None, max_num_match_pairs=1250000, max_count=3000,
                                keep_merges=False, all_ranks=False, workers=1, make_context=None, share_merges=0, min_shared_combos=9,
                                steal_stitch_cells=4000000, steal_chunk_cells=2000000):
    """progressive_msa over `world` ranks; every rank calls it with the same arguments (its own ctx).  `group` must be a
    host-tensor (gloo) process group.  Rank 0 returns the result dict (root graph, paths, scale, …), the others None — or, with
    all_ranks, their own dict (root None, stats of the merges they ran).  workers > 1: a rank that owns a whole subtree runs its
    independent merges (and its share of the calibrations) side by side on that many contexts of its device, as progressive_msa does.
    share_merges = G > 1 (level 3): a merge whose children were built by different ranks is run by a MERGE GROUP of up to G of the
    node's ranks (cl_peer_api.cpp): both children go to every member, every member runs the merge, and the far pass of its affine
    chaining DP is divided between the members' devices by chain combination — peer stores into one another's memory, no collective.
    Inside a merge group the merge's STITCH subproblems are shared as well (north_star: "shard naturally across the GPUs ... for work-stealing only"): a batch of at
    least steal_stitch_cells DP cells (0 / None: never) goes through the context's stitch hook (cl_context_set_stitch_hook) — every member pulls chunks of
    steal_chunk_cells cells of the LPT-ordered subproblem list from the group's ONE atomic counter (cl_context_peer_steal: a word in the leader's device memory),
    aligns them and hands its pieces to the other members over the host group, so that every member goes on with the complete, identical alignment."""
    import torch
    order = leaves_of(tree)
    make_context = make_context or (lambda: capi.Context(getattr(ctx, "device", 0)))
    share_merges = int(share_merges) if hasattr(ctx, "peer_export") and world > 1 else 0
    handles, merge_number, share_note = None, {}, None
    if share_merges > 1:
        handles = [None] * world
        try:
            mine = ctx.peer_export()
        except capi.ClError:
            mine = None                  # (no IPC export on this system: every rank will see the gap and go on without merge groups)
        dist.all_gather_object(handles, mine, group=group)
        # once round ALL ranks before anything depends on it: peer stores, stream memory operations and their order between these devices.
        # Any rank that sees nothing within five seconds switches the whole job back to one rank per merge.
        ok = torch.ones(1, dtype=torch.int32)
        # a context that has been in groups before: its arrival words are never reset, so this job's epochs and self-test token start above the
        # highest any member has used (the library refuses anything else)
        marks = torch.zeros(2, dtype=torch.int64)
        if mine is not None:
            st = ctx.peer_stats() if hasattr(ctx, "peer_stats") else {}
            marks[0], marks[1] = st.get("epoch_mark", 0), st.get("selftest_mark", 0)
        dist.all_reduce(marks, op=dist.ReduceOp.MAX, group=group)
        epoch0, token = int(marks[0]), int(marks[1]) + 1
        if any(h is None for h in handles) or world > 8:      # (more than one node's worth of ranks: no self-test round all of them, no groups)
            ok[0] = 0
        elif world <= 8:
            try:
                ctx.peer_group(handles, rank, epoch0)
                ok[0] = 1 if ctx.peer_selftest(token) else 0
                ctx.peer_group([], 0, 0)
            except capi.ClError:
                ok[0] = 0
            if int(ok[0]) == 0:
                ctx = make_context()   # (the old context's stream may be stuck behind a wait that nothing will satisfy: leave it alone)
        dist.all_reduce(ok, op=dist.ReduceOp.MIN, group=group)
        share_note = "merge groups of up to %d ranks" % share_merges
        if int(ok[0]) == 0:
            share_merges = 0
            share_note = "merge groups OFF: the peer self-test failed on some rank"

        def number(t):   # internal nodes in post-order: the same on every rank, growing from a node to its ancestors (the epochs of cl_context_peer_group)
            if not isinstance(t, str):
                number(t[0]); number(t[1])
                merge_number[newick(t)] = len(merge_number) + 1
        number(tree)
    contexts = [ctx] + [make_context() for _ in range(max(1, int(workers)) - 1)]
    # level 1: leaf calibrations, round-robin; the scales meet by a SUM all-reduce of a vector that is zero except at the
    # rank's own leaves (x + 0.0 is exact), and every rank takes the mean in leaf order like the reference (src/core.cpp:169-173)
    leaves = {}
    mine = torch.zeros(len(order), dtype=torch.float64)
    own = [(i, nm) for i, nm in enumerate(order) if i % world == rank]
    for i, nm in own:
        leaves[nm] = capi.leaf_graph(sequences[nm])
    for (i, nm), sc in zip(own, _in_parallel(contexts, [lambda c, nm=nm: c.leaf_intrinsic_scale(leaves[nm], max_count=max_count, max_num_match_pairs=max_num_match_pairs)
                                                        for i, nm in own])):
        mine[i] = sc
    dist.all_reduce(mine, group=group)
    scales = mine.tolist()
    scale = sum(scales) / len(scales)
    stats = dict(match_ms=0.0, align_ms=0.0, fuse_ms=0.0, merges=0, graphs_received=0)
    if share_note:
        stats["merge_groups"] = share_note

    def merge(g1, g2, t, c=None):
        r = (c or ctx).merge(g1, g2, score_scale=scale, max_num_match_pairs=max_num_match_pairs, max_count=max_count)
        return record(r, g1, g2, t)

    def record(r, g1, g2, t):
        for k in ("match_ms", "align_ms", "fuse_ms"):
            stats[k] += r[k]
        stats["merges"] += 1
        al = r.get("align")
        if al is not None:
            stats.setdefault("per_merge", []).append(dict(merge=newick(t), paths1=_count_leaves(t[0]), paths2=_count_leaves(t[1]),
                                                          match_sets=r["n_match_sets"], match_ms=r["match_ms"], align_ms=r["align_ms"], fuse_ms=r["fuse_ms"],
                                                          **{k: al[k] for k in ("chain_ms", "partition_ms", "stitch_ms", "chain_device_ms", "chain_pair_evals",
                                                                                "chain_match_pairs", "chain_combinations")}))
            if keep_merges:
                stats.setdefault("kept", []).append(dict(merge=newick(t), graphs=(g1, g2), align=al, fused=r["fused"], paths=leaves_of(t)))
        return r["fused"]

    def solve_local(t):
        """a whole subtree on this rank: its merges in waves, every merge whose two children are there side by side on the contexts"""
        done, pending = {}, []

        def collect(u):
            if isinstance(u, str):
                done[u] = leaves[u] if u in leaves else capi.leaf_graph(sequences[u])
            else:
                collect(u[0]); collect(u[1]); pending.append(u)
        collect(t)
        while pending:
            ready = [u for u in pending if newick(u[0]) in done and newick(u[1]) in done]
            pending = [u for u in pending if not (newick(u[0]) in done and newick(u[1]) in done)]
            jobs = [lambda c, u=u: c.merge(done[newick(u[0])], done[newick(u[1])], score_scale=scale, max_num_match_pairs=max_num_match_pairs, max_count=max_count)
                    for u in ready]
            for u, r in zip(ready, _in_parallel(contexts, jobs)):
                done[newick(u)] = record(r, done[newick(u[0])], done[newick(u[1])], u)   # (stats in tree order of the wave)
        return done[newick(t)]

    def solve(t, ranks):
        """the subtree's graph on ranks[0], None elsewhere"""
        if rank not in ranks:
            return None
        if isinstance(t, str):
            if rank != ranks[0]:
                return None   # only the owner needs the leaf's graph
            return leaves[t] if t in leaves else capi.leaf_graph(sequences[t])
        if len(ranks) == 1:
            if len(contexts) == 1:
                return merge(solve(t[0], ranks), solve(t[1], ranks), t)
            return solve_local(t)
        left, right = split_ranks(t, ranks)
        g1, g2 = solve(t[0], left), solve(t[1], right)
        combos = _count_leaves(t[0]) * _count_leaves(t[1])
        # (worth it from about nine combinations on: a far launch over one or two combinations is bound by the latency of a single query's
        # search, not by how many queries it holds — 323 µs for one combination, ~900 µs for the 25 of a 5 + 5-path root — and every member
        # needs both children's graphs)
        if share_merges > 1 and min_shared_combos <= combos <= 64 and epoch0 + 16 * merge_number[newick(t)] + 16 < 4096:
            members = [left[0], right[0]] + [r for r in ranks if r not in (left[0], right[0])][:share_merges - 2]
            if rank not in members:
                return None
            # both children to every member (the two builders send; a graph is a few tens of MB of host arrays)
            if rank == left[0]:
                for m in members[1:]:
                    send_graph(g1, m, dist, group)
            else:
                g1 = recv_graph(left[0], dist, group)
            if rank == right[0]:
                for m in members:
                    if m != right[0]:
                        send_graph(g2, m, dist, group)
            else:
                g2 = recv_graph(right[0], dist, group)
            stats["graphs_received"] += (rank != left[0]) + (rank != right[0])
            ctx.peer_group([handles[m] for m in members], members.index(rank), epoch0 + 16 * merge_number[newick(t)])
            hooked = [0]
            if steal_stitch_cells and hasattr(ctx, "set_stitch_hook"):
                from . import dist as cd
                job0 = epoch0 + 16 * merge_number[newick(t)]     # (the steal counter's job numbers grow like the epochs; up to 16 hooked batches per merge)

                def hook(batch, params, members=members, job0=job0):
                    job = job0 + hooked[0]
                    hooked[0] += 1
                    idx, res, took = cd.stitch_by_stealing(ctx, batch, lambda: ctx.peer_steal(job), chunk_cells=steal_chunk_cells, params=params)
                    parts = cd.exchange_objects((idx, res.aln_off, res.pairs, res.score, res.route, res.num_pw), members, rank, dist, group)
                    st = stats.setdefault("stitch_stealing", dict(batches=0, chunks_taken=0, problems_taken=0, problems_all=0))
                    st["batches"] += 1; st["chunks_taken"] += len(took); st["problems_taken"] += len(idx); st["problems_all"] += batch.n_problems
                    return cd.assemble_results(parts, batch.n_problems)
                ctx.set_stitch_hook(hook, steal_stitch_cells)
            try:
                r = ctx.merge(g1, g2, score_scale=scale, max_num_match_pairs=max_num_match_pairs, max_count=max_count)
            finally:
                if steal_stitch_cells and hasattr(ctx, "set_stitch_hook"):
                    ctx.set_stitch_hook(None)
                ctx.peer_group([], 0, 0)
            stats["shared_merges"] = stats.get("shared_merges", 0) + 1
            if rank != left[0]:
                return None                      # (the members' results are the leader's, bit for bit; only the leader keeps and records it)
            return record(r, g1, g2, t)
        if rank == right[0]:
            send_graph(g2, left[0], dist, group)
        if rank != left[0]:
            return None
        g2 = recv_graph(right[0], dist, group)
        stats["graphs_received"] += 1
        return merge(g1, g2, t)

    try:
        root = solve(tree, list(range(world)))
    finally:
        for c in contexts[1:]:
            if hasattr(c, "close"):
                c.close()
    if rank != 0:
        return dict(root=None, paths=order, scale=scale, scales=scales, stats=stats) if all_ranks else None
    return dict(root=root, paths=order, scale=scale, scales=scales, stats=stats)
