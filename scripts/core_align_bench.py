"""Core::align (cl_core_align) on the 2 x 1 Mbp pair made in place (cl_find_matches + cl_leaf_intrinsic_scale): phase times"""
import os
import sys
import time

HERE = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, HERE)
from centrolign_amd import capi, synth  # noqa: E402


def main():
    seqs = synth.hor_sequences(7, 1000000, 2)
    g1, g2 = synth.base_graph_from_sequence(seqs[0]), synth.base_graph_from_sequence(seqs[1], sentinels=(7, 8))
    ctx = capi.Context(0)
    ms = ctx.find_matches(g1, g2)
    scale = sum(ctx.leaf_intrinsic_scale(g) for g in (g1, g2)) / 2
    for rep in range(3):
        t0 = time.perf_counter()
        al = ctx.core_align(g1, g2, ms, score_scale=scale)
        print("rep %d: %.3f s wall; chain %.0f ms, partition %.0f ms, stitch %.0f ms; %d anchors, %d aligned pairs" %
              (rep, time.perf_counter() - t0, al["chain_ms"], al["partition_ms"], al["stitch_ms"], len(al["walk_off"]) - 1, len(al["alignment"])), flush=True)


if __name__ == "__main__":
    main()
