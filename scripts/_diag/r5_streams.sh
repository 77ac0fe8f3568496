# captured C3 step with / without the weight-gradient stream and the second attention stream: ms/step and the step's timeline
mkdir -p gpurun_out/r5d
for cfg in "1 1" "0 1" "1 0" "0 0"; do
  set -- $cfg
  NRM_WGRAD_STREAM=$1 NRM_BRANCH_STREAMS=$2 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --graph > gpurun_out/r5d/w$1_b$2.json 2> gpurun_out/r5d/w$1_b$2.err
  python - <<PY
import json
d=json.loads(open("gpurun_out/r5d/w$1_b$2.json").read().strip().splitlines()[-1])
print("wgrad", $1, "branch", $2, d["ms_per_step"], d["launch_probe"]["t_eager_ms"], d["launch_probe"]["t_graph_ms"], flush=True)
PY
done
cd /tmp && export TMPDIR=/tmp
for cfg in "0 1"; do
  set -- $cfg
  NRM_WGRAD_STREAM=$1 NRM_BRANCH_STREAMS=$2 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r5d/tr -- python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 2 --no-cpu-baseline --graph > /dev/null 2>&1
  f=$(find $GRAFT_REPO_ROOT/gpurun_out/r5d/tr -name "*kernel_trace.csv" | head -1)
  python3 $GRAFT_REPO_ROOT/scripts/_diag/timeline.py $f 10 full > $GRAFT_REPO_ROOT/gpurun_out/r5d/tl_w$1_b$2.txt 2>&1 || true
  rm -rf $GRAFT_REPO_ROOT/gpurun_out/r5d/tr
done
