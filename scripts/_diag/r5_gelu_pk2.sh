#!/bin/bash
out=gpurun_out/r5pk; mkdir -p $out
python -m pytest tests/test_gpu_attention.py -x -q -k "oracle or identical" > $out/tests2.log 2>&1; echo "tests rc=$?" | tee -a $out/tests2.log; tail -2 $out/tests2.log
B="NRM_ALLOW_STALE_LIB=1 NRM_HOTPATH_LIB=$PWD/scripts/_diag/libnrm_base.so"
for i in 1 2 3; do
  python scripts/_diag/fwd_probe.py 1024 30 50 400 12 | tee -a $out/fwd2.txt
  env $B python scripts/_diag/fwd_probe.py 1024 30 50 400 12 | tee -a $out/fwd2.txt
done
for i in 1 2 3; do
  python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-probe > $out/c3n_$i.json 2>$out/c3n_$i.err; python scripts/_diag/pr.py $out/c3n_$i.json
  env $B python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-probe > $out/c3b_$i.json 2>$out/c3b_$i.err; python scripts/_diag/pr.py $out/c3b_$i.json
done
