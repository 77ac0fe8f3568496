# hipGraph replay of the captured step with 2 / 4 (default) / 6 / 8 execution queues (DEBUG_HIP_FORCE_GRAPH_QUEUES): ms/step per workload
mkdir -p gpurun_out/r5b
for q in 4 2 6 8; do
  for w in C3-large C2-small ref-default; do
    DEBUG_HIP_FORCE_GRAPH_QUEUES=$q python bench.py --workload $w --steps 20 --warmup 5 --no-cpu-baseline --graph > gpurun_out/r5b/q${q}_$w.json 2> gpurun_out/r5b/q${q}_$w.err
    python - <<PY
import json
d=json.loads(open("gpurun_out/r5b/q${q}_$w.json").read().strip().splitlines()[-1])
print("queues", $q, "$w", d["ms_per_step"], d["launch_probe"]["t_eager_ms"], d["launch_probe"]["t_graph_ms"], flush=True)
PY
  done
done
cd /tmp && export TMPDIR=/tmp
for q in 8; do
  w=C3-large
  DEBUG_HIP_FORCE_GRAPH_QUEUES=$q rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r5b/tr -- python3 $GRAFT_REPO_ROOT/bench.py --workload $w --steps 3 --warmup 2 --no-cpu-baseline --graph > /dev/null 2>&1
  f=$(find $GRAFT_REPO_ROOT/gpurun_out/r5b/tr -name "*kernel_trace.csv" | head -1)
  python3 $GRAFT_REPO_ROOT/scripts/_diag/timeline.py $f 10 full > $GRAFT_REPO_ROOT/gpurun_out/r5b/tl_q${q}_$w.txt 2>&1 || true
  rm -rf $GRAFT_REPO_ROOT/gpurun_out/r5b/tr
done
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_fullsize_step.py tests/test_gpu_attention.py -x -q -k "full_size or default_dispatch" > gpurun_out/r5b/tests.log 2>&1; tail -5 gpurun_out/r5b/tests.log
