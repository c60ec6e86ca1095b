# deep (fused, bounded order) vs wide launch shape by call size on three kinds of content: where the size thresholds of launch_bc7 should sit
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/thresholds
for c in I3opaque baboon; do
  timeout 200 python tools/bc7_path_probe.py slow,alpha_slow bc7 $c 2>&1 | grep -v amdgpu
done | tee gpurun_out/thresholds/bc7_path_probe_by_content.txt
