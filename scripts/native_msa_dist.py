"""The progressive MSA of scripts/native_msa.py over several ranks (torchrun, one process per GPU; ranks beyond the GPU count share
devices): leaf calibrations round-robin, sibling subtrees on different ranks, fused graphs handed up the tree over a gloo group
(centrolign_amd/msa.progressive_msa_distributed).  Rank 0 checks the GFA against the serial driver's and reports both times."""
import hashlib
import os
import sys
import time

HERE = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, HERE)
from centrolign_amd import capi, msa, synth  # noqa: E402


def main():
    import torch.distributed as dist
    n, length, budget, seed = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    ctx = capi.Context(int(os.environ.get("LOCAL_RANK", "0")) % max(1, int(capi.load_library().cl_device_count())))
    seqs = synth.hor_sequences(seed, length, n, seq_div=0.01, hor_div=0.03, indel_hor=2)
    names = ["seq%d" % i for i in range(n)]
    tree = msa.balanced_tree(names)
    ctx.find_matches(capi.leaf_graph("ACGTACGTAC"), capi.leaf_graph("ACGTTCGTAC"))
    dist.barrier()
    t0 = time.perf_counter()
    r = msa.progressive_msa_distributed(ctx, dict(zip(names, seqs)), tree, dist, rank, world, max_num_match_pairs=budget)
    dist.barrier()
    t1 = time.perf_counter()
    if rank == 0:
        gfa = capi.write_gfa(r["root"], r["paths"])
        print("%d ranks: %.2f s; rank 0 did %d of %d merges, received %d graphs; GFA sha256 %s" %
              (world, t1 - t0, r["stats"]["merges"], n - 1, r["stats"]["graphs_received"], hashlib.sha256(gfa).hexdigest()[:16]), flush=True)
        t0 = time.perf_counter()
        s = msa.progressive_msa(ctx, dict(zip(names, seqs)), tree, max_num_match_pairs=budget)
        t1 = time.perf_counter()
        same = capi.write_gfa(s["root"], s["paths"]) == gfa
        print("serial on rank 0: %.2f s; GFA identical: %s" % (t1 - t0, same), flush=True)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
