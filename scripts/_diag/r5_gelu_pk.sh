#!/bin/bash
# packed-math GELU (gelu4 / gelu_dot4 / gelu_grad4_times in common.hpp) against the scalar evaluation (libnrm_base.so = the sources before it)
out=gpurun_out/r5pk; mkdir -p $out
python -m pytest tests/test_gpu_attention.py tests/test_gpu_dense.py tests/test_gpu_model.py -x -q > $out/tests.log 2>&1; echo "tests rc=$?" | tee -a $out/tests.log; tail -3 $out/tests.log
B="NRM_ALLOW_STALE_LIB=1 NRM_HOTPATH_LIB=$PWD/scripts/_diag/libnrm_base.so"
for shape in "1024 30 50 400" "256 64 128 768" "256 15 200 64" "256 20 10 256"; do
  for i in 1 2; do
    python scripts/_diag/fwd_probe.py $shape | tee -a $out/fwd.txt
    env $B python scripts/_diag/fwd_probe.py $shape | tee -a $out/fwd.txt
  done
done
for w in C3-large C2-small ref-default; do
  for i in 1 2; do
    python bench.py --workload $w --steps 20 --warmup 5 --no-cpu-baseline --no-probe > $out/${w}_new_$i.json 2>$out/${w}_new_$i.err; python scripts/_diag/pr.py $out/${w}_new_$i.json
    env $B python bench.py --workload $w --steps 20 --warmup 5 --no-cpu-baseline --no-probe > $out/${w}_base_$i.json 2>$out/${w}_base_$i.err; python scripts/_diag/pr.py $out/${w}_base_$i.json
  done
done
