#!/bin/bash
# dW_p-only pass with one accumulator set (NRM_DW_DIRECT, default on) against the two-set E-form, parity first, then C3 / C5 / C2-fp32
out=gpurun_out/r5dw; mkdir -p $out
python -m pytest tests/test_gpu_attention.py -x -q -k "direct_dw or dp_walk or skips or oracle" > $out/tests.log 2>&1; echo "tests rc=$?" | tee -a $out/tests.log
tail -4 $out/tests.log
grep -q "rc=0" $out/tests.log || exit 1
for shape in "1024 30 50 400" "256 64 128 768" "512 30 32 256" "256 15 200 64"; do
  for m in 0 1; do NRM_DW_DIRECT=$m python scripts/_diag/dp_probe.py $shape | tee -a $out/probe.txt; done
done
for i in 1 2; do
  for m in 0 1; do
    NRM_DW_DIRECT=$m python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-probe > $out/c3_dw${m}_$i.json 2> $out/c3_dw${m}_$i.err || { echo "bench failed"; exit 1; }
    python -c "
import json;d=json.loads(open('$out/c3_dw${m}_$i.json').read().strip().splitlines()[-1]);print('dw_direct=$m', d['ms_per_step'], d['value'], d['roofline']['step']['frac'])"
  done
done
