#!/bin/bash
# packed-math dz pass (gelu_dz4) against the scalar evaluation (libnrm_base.so)
out=gpurun_out/r5pk; mkdir -p $out
python -m pytest tests/test_gpu_attention.py -x -q -k "dz or oracle" > $out/tests3.log 2>&1; echo "tests rc=$?" | tee -a $out/tests3.log; tail -2 $out/tests3.log
B="NRM_ALLOW_STALE_LIB=1 NRM_HOTPATH_LIB=$PWD/scripts/_diag/libnrm_base.so"
for shape in "1024 30 50 400" "256 64 128 768" "512 30 32 256" "256 15 200 64"; do
  for i in 1 2; do
    echo "new $shape" | tee -a $out/dz.txt; python scripts/_diag/dz_bench.py $shape 2>&1 | tail -3 | tee -a $out/dz.txt
    echo "base $shape" | tee -a $out/dz.txt; env $B python scripts/_diag/dz_bench.py $shape 2>&1 | tail -3 | tee -a $out/dz.txt
  done
done
for i in 1 2; do
  python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-probe > $out/c3dz_n_$i.json 2>$out/c3dz_n_$i.err; python scripts/_diag/pr.py $out/c3dz_n_$i.json
  env $B python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-probe > $out/c3dz_b_$i.json 2>$out/c3dz_b_$i.err; python scripts/_diag/pr.py $out/c3dz_b_$i.json
done
