#!/bin/bash
# direct dW_p pass: operand sets (reduction steps requested ahead) 2 / 3 / 4 and wave-task counts, C3 and C5
out=gpurun_out/r5dw; mkdir -p $out
for w in 6144 3072 9216 12288; do
  echo "NRM_BT_WAVES=$w" | tee -a $out/sets.txt
  NRM_BT_WAVES=$w python scripts/_diag/dp_probe.py | tee -a $out/sets.txt
  for v in dws3 dws4; do NRM_BT_WAVES=$w NRM_HOTPATH_LIB=$PWD/scripts/_diag/libnrm_$v.so python scripts/_diag/dp_probe.py | tee -a $out/sets.txt; done
done
for v in product dws3 dws4; do
  if [ $v = product ]; then python scripts/_diag/dp_probe.py 256 64 128 768 | tee -a $out/sets.txt; else NRM_HOTPATH_LIB=$PWD/scripts/_diag/libnrm_$v.so python scripts/_diag/dp_probe.py 256 64 128 768 | tee -a $out/sets.txt; fi
done
