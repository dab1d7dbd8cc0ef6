import sys, os, subprocess; sys.path.insert(0,'/root/repo')
import squid_amd
os.makedirs('/tmp/s', exist_ok=True)
cfg = sys.argv[1] if len(sys.argv) > 1 else 'C2'
extra = sys.argv[2:]
pre='/tmp/s/'+cfg+'_'.join(extra).replace('-','')
if not os.path.exists(pre+'.bam'):
    subprocess.check_call(['/root/repo/build/gen_synth_bam','--config',cfg,'--out',pre,'--threads','16',*extra], stdout=subprocess.DEVNULL)
with squid_amd.Context() as ctx:
    ctx.load(pre+'.bam',pre+'.chim.bam')
    ctx.build_graph(); ctx.order()
    print({k:round(v['ms'],3) for k,v in ctx.timing().items() if 'order' in k or 'mincut' in k})
