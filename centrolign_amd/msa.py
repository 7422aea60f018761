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


def progressive_msa(ctx, sequences, tree, max_num_match_pairs=1250000, max_count=3000):
    """sequences: {name: str}.  Returns dict(root BaseGraph, paths [names in path order], alignment of the root merge, scale,
    scales, stats)"""
    order = leaves_of(tree)
    leaves = {nm: capi.leaf_graph(sequences[nm]) for nm in order}
    scales = [ctx.leaf_intrinsic_scale(leaves[nm], max_count=max_count, max_num_match_pairs=max_num_match_pairs) for nm in order]
    scale = sum(scales) / len(scales)                    # ScoreFunction::score_scale (src/core.cpp:169-184)
    stats = dict(match_ms=0.0, align_ms=0.0, fuse_ms=0.0, merges=0)
    last = {}

    def solve(t):
        if isinstance(t, str):
            return leaves[t], [t]
        (g1, p1), (g2, p2) = solve(t[0]), solve(t[1])
        r = ctx.merge(g1, g2, score_scale=scale, max_num_match_pairs=max_num_match_pairs, max_count=max_count)
        for k in ("match_ms", "align_ms", "fuse_ms"):
            stats[k] += r[k]
        stats["merges"] += 1
        last["alignment"], last["graphs"] = r["alignment"], (g1, g2)
        return r["fused"], p1 + p2
    root, paths = solve(tree)
    return dict(root=root, paths=paths, alignment=last.get("alignment"), root_inputs=last.get("graphs"), scale=scale, scales=scales,
                stats=stats, leaves=leaves)


def output_text(result):
    """what the CLI prints: the explicit CIGAR of a pairwise run, the GFA of an MSA (src/main.cpp)"""
    if len(result["paths"]) == 2:
        g1, g2 = result["root_inputs"]
        return capi.explicit_cigar(g1, g2, result["alignment"])
    return capi.write_gfa(result["root"], result["paths"])
