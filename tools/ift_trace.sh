export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; out=$R/gpurun_out/ift_tr2; mkdir -p $out
cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $out -o x -- python $R/tools/ift_probe.py 4 200 > /dev/null 2>&1
python - <<'PY'
import csv, os
for r in csv.DictReader(open(os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/ift_tr2/x_kernel_stats.csv")):
    print(r['Name'][:70], r['Calls'], r['AverageNs'], r['MinNs'])
PY
find $out -name "*.db" -delete
cd $R && python - <<'PY'
import sys, os, json
sys.path[:0] = [os.getcwd(), os.path.join(os.getcwd(), "robotdynamics.jl_amd"), os.path.join(os.getcwd(), "tools")]
import numpy as np, rdhip as RD, bench_configs as BC
rng = np.random.default_rng(5)
BC.rollout_case("satellite ImplicitMidpoint rollout fp64 4096 traj x 101 knots", RD.Satellite(), 4096, 101, RD.ImplicitMidpoint, np.float64, 5, rng)
BC.rollout_case("satellite ImplicitMidpoint rollout fp64 512 traj x 256 knots", RD.Satellite(), 512, 256, RD.ImplicitMidpoint, np.float64, 5, rng)
PY
