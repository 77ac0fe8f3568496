#!/bin/bash
# pair form of the direct dW_p pass (H % 4 == 2: no padded reduction step), parity first, then C3 per launch and per step
out=gpurun_out/r5dw; mkdir -p $out
python -m pytest tests/test_gpu_attention.py -x -q -k "direct_dw or dp_walk or skips or oracle" > $out/tests_pair.log 2>&1; echo "tests rc=$?" | tee -a $out/tests_pair.log
tail -4 $out/tests_pair.log
grep -q "rc=0" $out/tests_pair.log || exit 1
for i in 1 2 3; do
  for m in 1 0; do NRM_DW_PAIR=$m python scripts/_diag/dp_probe.py | sed "s/^/pair=$m /" | tee -a $out/pair.txt; done
done
for i in 1 2 3; do
  for m in 1 0; do
    NRM_DW_PAIR=$m python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-probe > $out/c3_pair${m}_$i.json 2> $out/c3_pair${m}_$i.err; python scripts/_diag/pr.py $out/c3_pair${m}_$i.json | sed "s/^/pair=$m /"
  done
done
