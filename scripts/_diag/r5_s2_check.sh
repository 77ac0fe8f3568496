#!/bin/bash
# full -m gpu suite, then the C3 line twice and the other shapes once (no CPU baseline, no probe)   usage: r5_s2_check.sh <outdir-tag>
out=gpurun_out/${1:-s2c}; mkdir -p $out
python -m pytest tests -m gpu -x -q > $out/gpu_tests.log 2>&1; echo "tests rc=$?" >> $out/gpu_tests.log; tail -3 $out/gpu_tests.log
grep -q "rc=0" $out/gpu_tests.log || exit 1
for i in 1 2; do python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-probe > $out/c3_$i.json 2> $out/c3_$i.err; python scripts/_diag/pr.py $out/c3_$i.json; done
for w in C5-long C2-small ref-default C1-demo; do python bench.py --workload $w --steps 10 --warmup 3 --no-cpu-baseline --no-probe > $out/$w.json 2> $out/$w.err; python scripts/_diag/pr.py $out/$w.json; done
