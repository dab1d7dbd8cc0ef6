"""Helpers around the CPU oracle (test infrastructure): run it with --dump and parse the stage files."""
import subprocess
from pathlib import Path


def run_oracle(build_dir, prefix, outdir, *flags, check=True):
    outdir = Path(outdir)
    dump = outdir / "dump"
    dump.mkdir(parents=True, exist_ok=True)
    cmd = [str(Path(build_dir) / "squid_oracle"), "-b", f"{prefix}.bam", "-c", f"{prefix}.chim.bam", "-o", str(outdir / "oracle"), "--dump", str(dump), *flags]
    rc = subprocess.call(cmd, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL if not check else None)
    if check and rc:
        raise subprocess.CalledProcessError(rc, cmd)
    return outdir / "oracle_sv.txt", dump


def read_nodes(path):
    rows = []
    for line in Path(path).read_text().splitlines():
        if line.startswith("#"):
            continue
        f = line.split("\t")
        rows.append((int(f[0]), int(f[1]), int(f[2]), int(f[3]), float.fromhex(f[4])) + ((int(f[5]),) if len(f) > 5 else ()))
    return rows


def read_edges(path):
    rows = []
    for line in Path(path).read_text().splitlines():
        if line.startswith("#"):
            continue
        rows.append(tuple(int(x) for x in line.split("\t")))
    return rows


def read_orders(path):
    out = []
    for line in Path(path).read_text().splitlines():
        if line.startswith("#"):
            continue
        out.append([int(x) for x in line.split("\t")[1].split(",")])
    return out


def read_breakpoints(path):
    out = []
    for line in Path(path).read_text().splitlines():
        if line.startswith("#"):
            continue
        f = line.split("\t")
        bps = []
        for item in f[4:]:
            bp, sup = item.split(":")
            b1, b2 = bp.split(",")
            s1, s2 = sup.split(",")
            bps.append((-1 if b1 == "-" else int(b1), -1 if b2 == "-" else int(b2), int(s1), int(s2)))
        out.append(bps)
    return out
