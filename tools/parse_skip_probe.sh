CFG=${1:-C3}
mkdir -p gpurun_out
python - "$CFG" <<'PY' > gpurun_out/ps_$CFG.log 2>&1
import os, subprocess, sys, tempfile, time
from pathlib import Path
ROOT = Path(os.getcwd())
sys.path.insert(0, str(ROOT))
import squid_amd
cfg = sys.argv[1]
with tempfile.TemporaryDirectory() as td:
    pre = Path(td) / cfg
    subprocess.check_call([str(ROOT / "build" / "gen_synth_bam"), "--config", cfg, "--out", str(pre), "--threads", "32"], stdout=subprocess.DEVNULL)
    for skip in (0, 1, 2, 4, 8, 16, 31):
        os.environ["SQUID_PARSE_SKIP"] = str(skip)
        with squid_amd.Context() as ctx:
            t0 = time.time(); ctx.load(f"{pre}.bam", f"{pre}.chim.bam", threads=16); dt = time.time() - t0
            tt = ctx.timing()
            print("skip", skip, f"load {dt*1e3:.0f} ms", {k: round(v['ms'], 1) for k, v in tt.items() if 'parse' in k}, flush=True)
PY
