#!/bin/bash
out=gpurun_out/r5dw; mkdir -p $out
python scripts/_diag/dp_probe.py | tee -a $out/sets2.txt
for v in dws6w2 dws8w2; do for w in 6144 4096; do echo "$v NRM_BT_WAVES=$w"; NRM_BT_WAVES=$w NRM_HOTPATH_LIB=$PWD/scripts/_diag/libnrm_$v.so python scripts/_diag/dp_probe.py | tee -a $out/sets2.txt; done; done
python scripts/_diag/dp_probe.py | tee -a $out/sets2.txt
