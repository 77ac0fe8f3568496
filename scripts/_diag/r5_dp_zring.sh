#!/bin/bash
# dP walk: stages of the dz ring (dz rows requested ZR - 1 chunks ahead; 2 = the first version), parity first
out=gpurun_out/r5dp; mkdir -p $out
python -m pytest tests/test_gpu_attention.py -x -q -k "dp_walk or direct_dw" > $out/tests_zr.log 2>&1; echo "tests rc=$?" | tee -a $out/tests_zr.log; tail -2 $out/tests_zr.log
grep -q "rc=0" $out/tests_zr.log || exit 1
for v in zr2 zr4 zr6; do
  NRM_HOTPATH_LIB=$PWD/scripts/_diag/libnrm_$v.so python -m pytest tests/test_gpu_attention.py -x -q -k "dp_walk" > $out/tests_$v.log 2>&1; echo "$v tests rc=$?"
done
for i in 1 2; do
  python scripts/_diag/dp_probe.py | tee -a $out/zring.txt
  for v in zr2 zr4 zr6; do NRM_HOTPATH_LIB=$PWD/scripts/_diag/libnrm_$v.so python scripts/_diag/dp_probe.py | tee -a $out/zring.txt; done
done
python scripts/_diag/dp_probe.py 256 64 128 768 | tee -a $out/zring.txt
for v in zr2 zr4; do NRM_HOTPATH_LIB=$PWD/scripts/_diag/libnrm_$v.so python scripts/_diag/dp_probe.py 256 64 128 768 | tee -a $out/zring.txt; done
