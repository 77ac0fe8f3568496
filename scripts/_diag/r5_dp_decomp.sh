#!/bin/bash
# the dP walk kernel decomposed (timing-only builds, NRM_DIAG_DP bits: 1 no dz DMA after a step's first chunk, 2 no epilogue, 4 no W DMA
# after the first) and the dP form against the E-form on the other shapes
out=gpurun_out/r5dp; mkdir -p $out
python scripts/_diag/dp_probe.py > $out/decomp.txt
for v in dp1 dp2 dp4 dp7; do
  NRM_ALLOW_DIAG_LIB=1 NRM_HOTPATH_LIB=$PWD/scripts/_diag/libnrm_$v.so python scripts/_diag/dp_probe.py >> $out/decomp.txt
done
python scripts/_diag/dp_probe.py >> $out/decomp.txt
for g in 512 1024 1536 3072; do echo "grid $g" >> $out/decomp.txt; NRM_DP_GRID=$g python scripts/_diag/dp_probe.py >> $out/decomp.txt; done
for shape in "256 64 128 768" "512 30 32 256" "256 15 200 64" "1024 30 50 400"; do
  NRM_BWD_DP=1 python scripts/_diag/dp_probe.py $shape >> $out/decomp.txt
  NRM_BWD_DP=0 python scripts/_diag/dp_probe.py $shape >> $out/decomp.txt
done
cat $out/decomp.txt
