#!/bin/bash
# what the driver runs at round end: the -m gpu suite, smoke(), the default bench line
out=gpurun_out/final; mkdir -p $out
python -m pytest tests -m gpu -x -q > $out/gpu_tests.log 2>&1; echo "tests rc=$?" >> $out/gpu_tests.log; tail -3 $out/gpu_tests.log
python -c "import __graft_entry__ as g; g.smoke()" > $out/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $out/smoke.log
( time python bench.py --gpus 1 --steps 20 --warmup 5 > $out/bench.json 2> $out/bench.err ) 2> $out/bench.time; echo "bench rc=$?"; tail -3 $out/bench.time
python scripts/_diag/pr.py $out/bench.json
