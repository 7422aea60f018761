"""GPU suite: drop-in demonstration on real reference objects.  oracle/_ref/adapter_demo (built in the build
container from oracle/adapter_demo.cpp + the unmodified reference, shipped prebuilt to the GPU box) runs the
reference pipeline and stitches every merge twice — reference CPU loop vs include/centrolign_amd/stitch_adapter.hpp
-> C ABI -> MI355X — and compares the stitched Alignments."""
import os
import subprocess

import pytest

from centrolign_amd import synth

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DEMO = os.path.join(ROOT, "oracle", "_ref", "adapter_demo")


@pytest.mark.skipif(not os.path.exists(DEMO), reason="oracle/_ref/adapter_demo not built (needs the reference sources)")
@pytest.mark.parametrize("seed,length,n,max_pairs", [(21, 60000, 2, 60000), (22, 30000, 4, 40000)])
def test_reference_stitch_loop_vs_gpu(tmp_path, seed, length, n, max_pairs):
    seqs = synth.hor_sequences(seed, length, n, seq_div=0.01, hor_div=0.03, indel_hor=2)
    fa = str(tmp_path / "in.fa")
    synth.write_fasta(fa, seqs)
    nwk = "-"
    if n == 4:
        nwk = str(tmp_path / "tree.nwk")
        with open(nwk, "w") as f:
            f.write("((seq0,seq1),(seq2,seq3));")
    p = subprocess.run([DEMO, fa, nwk, str(max_pairs)], capture_output=True, text=True, timeout=600)
    print(p.stdout)
    print(p.stderr)
    assert p.returncode == 0, p.stdout + p.stderr
    assert "DROP-IN OK" in p.stdout
    assert p.stdout.count("== the reference") == (1 if n == 2 else 3)


@pytest.mark.skipif(not os.path.exists(DEMO), reason="oracle/_ref/adapter_demo not built (needs the reference sources)")
@pytest.mark.parametrize("seed,length,n,max_pairs", [(23, 60000, 2, 60000), (24, 30000, 4, 40000), (25, 40000, 4, 200000)])
def test_reference_core_align_vs_gpu(tmp_path, seed, length, n, max_pairs):
    """the same one level up: Core::align (anchor chain -> partition -> despecify -> stitch) of the unmodified reference
    against include/centrolign_amd/core_adapter.hpp -> cl_core_align, from the reference's own match sets, merge after
    merge of a fresh FASTA (later merges depend on the earlier GPU results)"""
    seqs = synth.hor_sequences(seed, length, n, seq_div=0.01, hor_div=0.03, indel_hor=2)
    fa = str(tmp_path / "in.fa")
    synth.write_fasta(fa, seqs)
    nwk = "-"
    if n == 4:
        nwk = str(tmp_path / "tree.nwk")
        with open(nwk, "w") as f:
            f.write("((seq0,seq1),(seq2,seq3));")
    p = subprocess.run([DEMO, fa, nwk, str(max_pairs), "core"], capture_output=True, text=True, timeout=900)
    print(p.stdout)
    print(p.stderr)
    assert p.returncode == 0, p.stdout + p.stderr
    assert "DROP-IN OK" in p.stdout
    assert p.stdout.count("GPU Core::align == the reference") == (1 if n == 2 else 3)


@pytest.mark.skipif(not os.path.exists(DEMO), reason="oracle/_ref/adapter_demo not built (needs the reference sources)")
@pytest.mark.parametrize("seed,length,n,max_pairs", [(26, 30000, 2, 50000), (27, 12000, 4, 30000)])
def test_reference_signatures_at_the_seams(tmp_path, seed, length, n, max_pairs):
    """include/centrolign_amd/seam_wrappers.hpp: the reference's own signatures — Anchorer::anchor_chain (with and without masked matches
    and an overriding scale), po_poa<NumPW>, Stitcher::internal_stitch — called next to the reference's functions on the same objects"""
    seqs = synth.hor_sequences(seed, length, n, seq_div=0.01, hor_div=0.03, indel_hor=2)
    fa = str(tmp_path / "in.fa")
    synth.write_fasta(fa, seqs)
    nwk = "-"
    if n == 4:
        nwk = str(tmp_path / "tree.nwk")
        with open(nwk, "w") as f:
            f.write("((seq0,seq1),(seq2,seq3));")
    p = subprocess.run([DEMO, fa, nwk, str(max_pairs), "seams"], capture_output=True, text=True, timeout=900)
    print(p.stdout)
    print(p.stderr)
    assert p.returncode == 0, p.stdout + p.stderr
    assert "DROP-IN OK" in p.stdout and "!=" not in p.stdout
    assert p.stdout.count("masked anchor_chain wrapper == the reference") == n
    assert p.stdout.count("internal_stitch wrapper ==") == n
    assert p.stdout.count("anchor_chain wrapper == the reference (") >= (1 if n == 2 else 3)
    assert p.stdout.count("po_poa<1|2|3> wrapper == on") == (1 if n == 2 else 3)
    assert p.stdout.count("masked + split anchor_chain wrapper == the reference") == (1 if n == 2 else 3)


@pytest.mark.skipif(not os.path.exists(DEMO), reason="oracle/_ref/adapter_demo not built (needs the reference sources)")
@pytest.mark.parametrize("seed,length,n,max_pairs", [(31, 40000, 2, 60000), (32, 20000, 5, 40000)])
def test_core_facade_prints_what_the_references_core_prints(tmp_path, seed, length, n, max_pairs):
    """include/centrolign_amd/core_facade.hpp: centrolign_amd::Core / Execution with the reference's member names (core.hpp:30-103,
    execution.hpp:33-120) — Core(fasta, tree), tunables, execute(), root_subproblem() — beside the unmodified reference's Core on the
    same files: the same CIGAR (two sequences) / GFA (five sequences, an unbalanced guide tree with an internal node of three leaves)"""
    seqs = synth.hor_sequences(seed, length, n, seq_div=0.01, hor_div=0.03, indel_hor=2)
    fa = str(tmp_path / "in.fa")
    synth.write_fasta(fa, seqs)
    nwk = "-"
    if n == 5:
        nwk = str(tmp_path / "tree.nwk")
        with open(nwk, "w") as f:
            f.write("((seq0,seq1),((seq2,seq3),seq4));")
    p = subprocess.run([DEMO, fa, nwk, str(max_pairs), "facade"], capture_output=True, text=True, timeout=900)
    print(p.stdout)
    print(p.stderr)
    assert p.returncode == 0, p.stdout + p.stderr
    assert "DROP-IN OK" in p.stdout and "== the reference's Core" in p.stdout
