# what the forward attention kernel's time is made of: timing-only builds (NRM_DIAG_FWD bits: 1 no accumulator-init loads, 2 no K-chunk
# DMA after the first, 4 no z store, 8 no GELU / fc2 dot) against the product, same box, C3 shape
mkdir -p gpurun_out/r5f
python scripts/_diag/fwd_probe.py > gpurun_out/r5f/fwd_decomp.txt
for v in fwd1 fwd2 fwd4 fwd8 fwd12 fwd15; do
  NRM_ALLOW_DIAG_LIB=1 NRM_HOTPATH_LIB=$PWD/scripts/_diag/libnrm_$v.so python scripts/_diag/fwd_probe.py >> gpurun_out/r5f/fwd_decomp.txt
done
python scripts/_diag/fwd_probe.py >> gpurun_out/r5f/fwd_decomp.txt
cat gpurun_out/r5f/fwd_decomp.txt
