#!/bin/bash
# Does the 16-byte-store hazard of csrc/common.hpp (store_b128_guarded) show on this GPU, and are the wait states alone enough?
# Runs the parity tests that caught it in round 3 (walk forward: bf16 attention shapes; resident-X GEMM: bf16 dense shapes) against
# three builds: guard 0 (none), 1 (s_nop 3 only), 2 (product: s_nop 3 between scheduling barriers).
for g in 0 1; do
  NRM_ALLOW_DIAG_LIB=1 NRM_HOTPATH_LIB=scripts/_diag/libnrm_guard$g.so python -m pytest tests/test_gpu_attention.py tests/test_gpu_dense.py -m gpu -q -k "bf16 or dense or walk or full_size_c2" -p no:cacheprovider 2>&1 | tail -4 > gpurun_out/r4_guard$g.log
  echo "guard $g:"; tail -2 gpurun_out/r4_guard$g.log
done
python -m pytest tests/test_gpu_attention.py tests/test_gpu_dense.py -m gpu -q -k "bf16 or dense or walk or full_size_c2" -p no:cacheprovider 2>&1 | tail -2
