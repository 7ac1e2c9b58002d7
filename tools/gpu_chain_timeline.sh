cd /tmp; export TMPDIR=/tmp
cat > /tmp/one.py <<'PY'
import sys, numpy as np
sys.path.insert(0, "/root/repo")
from cmax_slam_amd import _lib, evaluator, synth
chain = int(sys.argv[1])
p = synth.config2()
fe = evaluator.FrontendEvaluator(p.W, p.H, p.lut)
fe.set_option(_lib.OPT_CHAIN_SOLVE, chain)
fe.set_packet(p.x, p.y, p.t_ns, p.t_ref_ns, p.fx, p.fy, p.cx, p.cy, p.batch, p.sigma, _lib.VARIANCE)
for _ in range(40):
    fe.setupProblemAndOptimize(np.zeros(3))
PY
for c in 1 0; do
  rocprofv3 --kernel-trace --output-format csv -d /root/repo/gpurun_out/tl_$c -o t -- python /tmp/one.py $c > /dev/null 2>&1
  python /root/repo/tools/chain_timeline.py /root/repo/gpurun_out/tl_$c 64 > /root/repo/gpurun_out/timeline_chain$c.txt 2>&1
  rm -rf /root/repo/gpurun_out/tl_$c
done
