#!/usr/bin/env python3
"""Generates the annotate fixtures: runs the REAL reference script /root/reference/utils/AnnotateSQUIDOutput.py (a Python
reference can be run in the authoring container; it cannot travel to the GPU box) on small synthetic GTF / _sv.txt inputs
written by this script, and commits inputs + expected outputs as data under tests/golden/annotate/.

    python3 tests/golden/annotate/make_annotate_golden.py

The reference builds `genes` with list(set(...)) (utils/AnnotateSQUIDOutput.py:239), whose order depends on Python's string
hash seed; the FusedGenes column is therefore compared as a multiset of pairs (tests/test_annotate.py), everything else
byte for byte."""
import os
import random
import subprocess
import sys
from pathlib import Path

HERE = Path(__file__).resolve().parent
REF = Path("/root/reference/utils/AnnotateSQUIDOutput.py")


def make_case(name, seed, nchr, ngenes, nsv, gene_key="gene_id", sym_key="gene_name", out_of_order_exons=False, exon_only=False):
    rng = random.Random(seed)
    chrs = [f"chr{i + 1}" for i in range(nchr)] + (["chrX"] if nchr > 2 else [])
    lines = ["# synthetic annotation for the annotate fixtures\n"]
    genes = []
    late = []
    for g in range(ngenes):
        c = rng.choice(chrs)
        start = rng.randrange(1000, 2_000_000)
        strand = rng.choice("+-")
        gid, gname = f"G{g:04d}", f"SYM{g}"
        ntx = rng.randrange(1, 4)
        gstart, gend = None, None
        for t in range(ntx):
            tid = f"T{g:04d}.{t}"
            pos = start + rng.randrange(0, 500)
            exons = []
            for _ in range(rng.randrange(1, 6)):
                ln = rng.randrange(80, 400)
                exons.append((pos, pos + ln))
                pos += ln + rng.randrange(200, 5000)
            ts, te = exons[0][0], exons[-1][1]
            gstart = ts if gstart is None else min(gstart, ts)
            gend = te if gend is None else max(gend, te)
            attr = f'{gene_key} "{gid}"; transcript_id "{tid}"; {sym_key} "{gname}";'
            last_tx = g == ngenes - 1 and t == ntx - 1
            # exon_only: some transcripts have no transcript row at all (known from their exon rows only), and the last transcript of
            # the file gets one of its exons late -- in the script that exon is appended through the table to the very object its loop
            # variable still refers to
            if not (exon_only and not last_tx and rng.random() < 0.2):
                lines.append(f"{c}\tsynth\ttranscript\t{ts}\t{te}\t.\t{strand}\t.\t{attr}\n")
            for k, (a, b) in enumerate(exons):
                row = f"{c}\tsynth\texon\t{a}\t{b}\t.\t{strand}\t.\t{attr} exon_number \"{k + 1}\";\n"
                if exon_only and last_tx and k == len(exons) - 1 and len(exons) > 1:
                    late.append(row)
                elif out_of_order_exons and rng.random() < 0.15:
                    late.append(row)  # exon rows that come after another transcript's record (the script's extraExons path)
                else:
                    lines.append(row)
        genes.append((c, gstart, gend, strand))
    rng.shuffle(late)
    lines += late
    (HERE / f"{name}.gtf").write_text("".join(lines))
    sv = ["# chrom1\tstart1\tend1\tchrom2\tstart2\tend2\tname\tscore\tstrand1\tstrand2\tnum_concordantfrag_bp1\tnum_concordantfrag_bp2\n"]
    for _ in range(nsv):
        def side():
            if rng.random() < 0.8:
                c, s, e, _ = rng.choice(genes)
                p = rng.randrange(s - 200, e + 200)
            else:
                c, p = rng.choice(chrs), rng.randrange(1000, 2_500_000)
            a, b = sorted((p, p + rng.randrange(50, 3000)))
            return c, a, b
        c1, a1, b1 = side()
        c2, a2, b2 = side()
        sv.append(f"{c1}\t{a1}\t{b1}\t{c2}\t{a2}\t{b2}\t.\t{rng.randrange(3, 90)}\t{rng.choice('+-')}\t{rng.choice('+-')}\t{rng.randrange(0, 50)}\t{rng.randrange(0, 50)}\n")
    (HERE / f"{name}_sv.txt").write_text("".join(sv))
    args = []
    if gene_key != "gene_id":
        args += ["--geneid", gene_key]
    if sym_key != "gene_name":
        args += ["--genesymbol", sym_key]
    env = dict(os.environ, PYTHONHASHSEED="0")
    subprocess.check_call([sys.executable, str(REF)] + args + [str(HERE / f"{name}.gtf"), str(HERE / f"{name}_sv.txt"), str(HERE / f"{name}_expected.txt")], env=env)
    (HERE / f"{name}.args").write_text(" ".join(args) + "\n")


if __name__ == "__main__":
    if not REF.exists():
        sys.exit("reference tree absent: the committed fixtures are used as they are")
    make_case("a1", 1, 3, 40, 60)
    make_case("a2", 2, 1, 120, 80)                       # one chromosome, dense: overlapping genes, many candidates per breakpoint
    make_case("a3", 3, 4, 60, 50, "gene", "symbol")      # --geneid / --genesymbol
    make_case("a4", 4, 3, 50, 60, out_of_order_exons=True)
    make_case("a5", 5, 2, 3, 10)                         # fewer genes than the 20-step walks of LocatePosition_generange
    make_case("a6", 6, 3, 50, 60, out_of_order_exons=True, exon_only=True)  # transcripts known from exon rows only + a late exon of the last transcript row
    print("fixtures written to", HERE)
