# timeline of an EAGER C3 step (real streams), final defaults
mkdir -p $GRAFT_REPO_ROOT/gpurun_out/r5p
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r5p/tr -- python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 2 --no-cpu-baseline --eager > /dev/null 2>&1
f=$(find $GRAFT_REPO_ROOT/gpurun_out/r5p/tr -name "*kernel_trace.csv" | head -1)
python3 $GRAFT_REPO_ROOT/scripts/_diag/timeline.py $f 12 full > $GRAFT_REPO_ROOT/gpurun_out/r5p/tl_eager_c3.txt 2>&1
rm -rf $GRAFT_REPO_ROOT/gpurun_out/r5p/tr
head -20 $GRAFT_REPO_ROOT/gpurun_out/r5p/tl_eager_c3.txt
cd $GRAFT_REPO_ROOT && python -m pytest tests/test_gpu_dp.py tests/test_gpu_fullsize_step.py tests/test_gpu_model.py -m gpu -x -q -k "bench or full_size or weight_gradient" > gpurun_out/r5p/t.log 2>&1; tail -3 gpurun_out/r5p/t.log
python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r5p/c3.json 2>gpurun_out/r5p/c3.err; python -c "
import json; d=json.loads(open('gpurun_out/r5p/c3.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['config']['launch'], d['launch_probe'])"
