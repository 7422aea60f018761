"""child of tests/test_gpu_merge_group.py: one rank of a two-rank MSA on ONE device (started under torch.distributed.run, gloo, before any
GPU call of its own): the root merge is run by both ranks as a merge group (centrolign_amd.msa.progressive_msa_distributed, share_merges).
usage: merge_group_child.py length budget      (ten sequences over the guide tree of BASELINE configs[2]: the root merge has 5 x 5 chain combinations)"""
import hashlib
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from centrolign_amd import capi, dist as cd, msa, synth  # noqa: E402


def main():
    length, budget = [int(x) for x in sys.argv[1:3]]
    rank, world, dist = cd.init_distributed("gloo")
    names, seqs, tree = synth.c3_workload(length)
    ctx = capi.Context(0)
    # (the root merge's stitch batch is small at this length: the thresholds are lowered so that the members do share it through the group's steal counter)
    r = msa.progressive_msa_distributed(ctx, seqs, tree, dist, rank, world, max_num_match_pairs=budget, all_ranks=True, share_merges=world,
                                        steal_stitch_cells=20000, steal_chunk_cells=30000)
    st = ctx.peer_stats()
    text = msa.output_text(r) if rank == 0 else b""
    ss = r["stats"].get("stitch_stealing", dict(batches=0, chunks_taken=0, problems_taken=0, problems_all=0))
    print("RANK %d shared_merges=%d shared_dps=%d far_launches=%d merged_blocks=%d sha=%s" %
          (rank, r["stats"].get("shared_merges", 0), st["shared_dps"], st["shared_far_launches"], st["merged_blocks"],
           hashlib.sha256(text).hexdigest() if rank == 0 else "-"), flush=True)
    print("STEAL %d batches=%d chunks=%d problems=%d of=%d steals=%d" % (rank, ss["batches"], ss["chunks_taken"], ss["problems_taken"], ss["problems_all"], st["steals"]), flush=True)
    dist.barrier()
    ctx.close()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
