cd $GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out/r02c; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for cfg in "64 slow" "64 basic" "8 slow" "256 slow"; do
  set -- $cfg
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/t -o t -- python $GRAFT_REPO_ROOT/tools/wide_trace_probe.py $1 $2 wide > /dev/null 2> $OUT/err.log
  echo "== rows=$1 $2"; find $OUT/t -name '*kernel_stats*.csv' | head -1 | xargs cat | cut -d, -f1-8 | grep -E "bc7|Name" | sed 's/void itw:://' | cut -c1-200
  rm -rf $OUT/t
done
cd $GRAFT_REPO_ROOT
for t in BC7_slow BC7_basic BC1 BC6H_slow; do
  python - <<PY
import sys,os,subprocess,tempfile
sys.path.insert(0,'.'); sys.path.insert(0,'intel-texture-works-plugin_amd')
from itw_amd import surfaces
t="$t"
p='/tmp/in_%s.raw'%t
(surfaces.hdr_smooth(4096,4096) if t.startswith('BC6H') else surfaces.ldr_smooth(4096,4096)).tofile(p)
for mode,w in (('st',1),('mt',8),('mt',64)):
  for extra in ([],['whole']):
    r=subprocess.run(['oracle/_ref/ref_threads_caller_gpu',mode,t,'4096','4096',p,'/tmp/out.bin']+extra,capture_output=True,text=True,env=dict(os.environ,ITW_REF_THREADS=str(w),ITW_REF_REPS='4'))
    print(r.stdout.strip() or r.stderr[-300:], flush=True)
PY
done
