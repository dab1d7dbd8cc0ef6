# GPU-box probe: host ingest pipeline vs SQUID_GPU_INFLATE=1 on one generator config (default C2)
CFG=${1:-C2}
mkdir -p gpurun_out
python - "$CFG" <<'PY' > gpurun_out/gi_$CFG.log 2>&1
import os, subprocess, sys, tempfile, time
from pathlib import Path
ROOT = Path(os.getcwd())
sys.path.insert(0, str(ROOT))
import squid_amd
cfg = sys.argv[1]
os.environ["SQUID_INGEST_TIMING"] = "1"
with tempfile.TemporaryDirectory() as td:
    pre = Path(td) / cfg
    subprocess.check_call([str(ROOT / "build" / "gen_synth_bam"), "--config", cfg, "--out", str(pre), "--threads", "32"], stdout=subprocess.DEVNULL)
    print("bam bytes", os.path.getsize(f"{pre}.bam"), flush=True)
    envs = [{"SQUID_GPU_INFLATE": "0"}, {}, {}]
    if cfg == "C2": envs.insert(1, {"SQUID_GPU_INFLATE": "1", "SQUID_INFLATE_CHECK": "1"})
    for env in envs:
        for k in ("SQUID_GPU_INFLATE", "SQUID_INFLATE_CHECK"): os.environ.pop(k, None)
        os.environ.update(env)
        with squid_amd.Context() as ctx:
            t0 = time.time(); ctx.load(f"{pre}.bam", f"{pre}.chim.bam", threads=16); dt = time.time() - t0
            n = ctx.counts()['n_concordant']
            print(env, f"load {dt*1e3:.1f} ms, {n/dt/1e6:.1f} M rec/s", ctx.counts(), flush=True)
            tt = ctx.timing()
            for k, v in tt.items():
                if "infl" in k or "rec_" in k or "parse" in k or "lz_" in k: print("   ", k, v, flush=True)
PY
