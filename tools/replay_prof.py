import sys; sys.path.insert(0,'/root/repo')
import squid_amd, subprocess, os
os.makedirs('/tmp/s', exist_ok=True)
subprocess.check_call(['/root/repo/build/gen_synth_bam','--config','C2','--out','/tmp/s/C2'], stdout=subprocess.DEVNULL)
with squid_amd.Context() as ctx:
    ctx.load('/tmp/s/C2.bam','/tmp/s/C2.chim.bam')
    for i in range(3):
        ctx.reset(); ctx.build_graph()
