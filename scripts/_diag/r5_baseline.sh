set -e
mkdir -p gpurun_out/r5a
python -m pytest tests/test_gpu_dp.py -x -q > gpurun_out/r5a/dp.log 2>&1 || { tail -30 gpurun_out/r5a/dp.log; exit 1; }
tail -3 gpurun_out/r5a/dp.log
python bench.py --steps 20 --warmup 5 > gpurun_out/r5a/c3.json 2> gpurun_out/r5a/c3.err
python bench.py --workload C2-small --steps 30 --warmup 5 --no-cpu-baseline > gpurun_out/r5a/c2.json 2> gpurun_out/r5a/c2.err
python bench.py --workload ref-default --steps 50 --warmup 5 --no-cpu-baseline > gpurun_out/r5a/ref.json 2> gpurun_out/r5a/ref.err
python bench.py --workload C1-demo --steps 50 --warmup 5 --no-cpu-baseline > gpurun_out/r5a/c1.json 2> gpurun_out/r5a/c1.err
cd /tmp && export TMPDIR=/tmp
for w in C3-large C2-small ref-default; do
  NRM_X=1 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r5a/tr_$w -- python3 $GRAFT_REPO_ROOT/bench.py --workload $w --steps 3 --warmup 2 --no-cpu-baseline --graph > /dev/null 2>&1
  f=$(find $GRAFT_REPO_ROOT/gpurun_out/r5a/tr_$w -name "*kernel_trace.csv" | head -1)
  python3 $GRAFT_REPO_ROOT/scripts/_diag/timeline.py $f 10 full > $GRAFT_REPO_ROOT/gpurun_out/r5a/tl_$w.txt 2>&1 || true
  rm -rf $GRAFT_REPO_ROOT/gpurun_out/r5a/tr_$w
done
echo ok
