"""The unmodified reference's -c output on some inputs depends on what the heap happens to hold: Bonder::trim_partition_ends (src/bonder.cpp:753-757) takes off
intervening_segments[interval.second] — one element PAST that vector when the trimmed interval ends at the last shared segment — and goes on with those three doubles.
This script shows it without touching the reference: the compiled CLI (oracle/_ref/ref_cli, built from /root/reference by oracle/Makefile) on the inputs
scripts/fuzz_msa.py found (tests/golden/reference_undefined_cases.json), once as it is and once per MALLOC_PERTURB_ value (glibc then fills freed and fresh heap memory
with that byte / its complement), and once through the recorded flow of oracle/ref_driver.cpp (another program round the same objects): the sha256 of the text it prints, beside the text this library prints for the same input (recorded on the GPU box).
usage (build container, CPU only; a case takes 1-3 minutes): python scripts/reference_undefined_read.py [name ...] [--json OUT]"""
import hashlib
import json
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from centrolign_amd import synth  # noqa: E402

REF_CLI = os.path.join(ROOT, "oracle", "_ref", "ref_cli")
FILLS = (None, "1", "2", "85", "170", "255")


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    out_path = sys.argv[sys.argv.index("--json") + 1] if "--json" in sys.argv else None
    if out_path in args:
        args.remove(out_path)
    cases = json.load(open(os.path.join(ROOT, "tests", "golden", "reference_undefined_cases.json")))
    rows = []
    for p in cases:
        if args and p["name"] not in args:
            continue
        seqs = synth.tandem_dup_sequences(p["seed"], p["length"], p["n"], p["dup"], carriers=p["carriers"], seq_div=p["seq_div"], hor_div=p["hor_div"])
        names = ["q%02d" % i for i in range(p["n"])]
        with tempfile.TemporaryDirectory() as d:
            synth.write_fasta(os.path.join(d, "in.fa"), seqs, names)
            open(os.path.join(d, "t.nwk"), "w").write(p["newick"] + "\n")
            over = "i:max_num_match_pairs=%d;b:cyclize_tandem_duplications=1;i:min_cyclizing_length=%d" % (p["budget"], p["min_cyclizing_length"])
            texts = {}
            for fill in FILLS + (None,):
                env = dict(os.environ)
                env.pop("MALLOC_PERTURB_", None)
                if fill:
                    env["MALLOC_PERTURB_"] = fill
                r = subprocess.run([REF_CLI, "in.fa", "t.nwk", "-", "out.txt", "0", "0", "0", over], cwd=d, capture_output=True, text=True, env=env)
                key = ("MALLOC_PERTURB_=" + fill) if fill else ("unset" if "unset" not in texts else "unset, again")
                texts[key] = hashlib.sha256(open(os.path.join(d, "out.txt"), "rb").read()).hexdigest()[:16] if r.returncode == 0 else "exit %d" % r.returncode
            # another program round the same unmodified objects: the recorded flow of oracle/ref_driver.cpp (its dumps make other allocations in between: another heap)
            code = "import sys; sys.path.insert(0, %r); from oracle import pyoracle as po; po.ref_cyclize_dump('in.fa', 't.nwk', 'd.bin', 'out2.txt', %r)" % (
                ROOT, "i:min_cyclizing_length=%d;i:max_num_match_pairs=%d" % (p["min_cyclizing_length"], p["budget"]))
            r = subprocess.run([sys.executable, "-c", code], cwd=d, capture_output=True, text=True)
            texts["the recorded flow (oracle/ref_driver.cpp)"] = hashlib.sha256(open(os.path.join(d, "out2.txt"), "rb").read()).hexdigest()[:16] if r.returncode == 0 else "exit %d" % r.returncode
        row = dict(name=p["name"], sequences=p["n"], length=p["length"], reference_text_by_heap_fill=texts, distinct_reference_texts=len(set(texts.values())),
                   library_text=p["library_text_sha256"][:16], library_text_is_one_of_the_references=p["library_text_sha256"][:16] in texts.values())
        rows.append(row)
        print(json.dumps(row), flush=True)
    if out_path:
        with open(out_path, "w") as f:
            json.dump(rows, f, indent=1)


if __name__ == "__main__":
    main()
