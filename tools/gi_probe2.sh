# GPU-box probe: SQUID_GPU_INFLATE=1 at several token-batch sizes
CFG=${1:-C3}
mkdir -p gpurun_out
python - "$CFG" <<'PY' > gpurun_out/gi2_$CFG.log 2>&1
import os, subprocess, sys, tempfile, time
from pathlib import Path
ROOT = Path(os.getcwd())
sys.path.insert(0, str(ROOT))
import squid_amd
cfg = sys.argv[1]
os.environ["SQUID_INGEST_TIMING"] = "1"
os.environ["SQUID_GPU_INFLATE"] = "1"
with tempfile.TemporaryDirectory() as td:
    pre = Path(td) / cfg
    subprocess.check_call([str(ROOT / "build" / "gen_synth_bam"), "--config", cfg, "--out", str(pre), "--threads", "32"], stdout=subprocess.DEVNULL)
    for cap in (720, 880, 1000, 600):
        os.environ["SQUID_TOK_CAP_MB"] = str(cap)
        t0 = time.time()
        with squid_amd.Context() as ctx:
            t1 = time.time(); ctx.load(f"{pre}.bam", f"{pre}.chim.bam", threads=16); dt = time.time() - t1
            n = ctx.counts()['n_concordant']
            print("cap MB", cap, f"load {dt*1e3:.1f} ms, {n/dt/1e6:.1f} M rec/s", flush=True)
            tt = ctx.timing()
            print("   ", {k: round(v['ms'], 1) for k, v in tt.items() if "infl" in k or "lz_" in k}, flush=True)
        print(f"   context lifetime {1e3*(time.time()-t0):.0f} ms", flush=True)
PY
