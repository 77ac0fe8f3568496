#!/bin/bash
# Diagnostic (round 5): the dt + dW_p pass as the software-pipelined kernel with a third accumulator set, one wave per SIMD
# (NRM_BT_PIPE=1), against the shipped serial-epilogue kernel at two waves per SIMD.  Parity first, then the C3 kernel table.
set -o pipefail
mkdir -p gpurun_out/r5bt
NRM_BT_PIPE=1 timeout -k 10 600 python -m pytest tests/test_gpu_attention.py tests/test_gpu_model.py -m gpu -x -q > gpurun_out/r5bt/tests_pipe.log 2>&1
echo "tests rc=$?"; tail -3 gpurun_out/r5bt/tests_pipe.log
for arm in base pipe2048 pipe1024 base2; do
  case $arm in
    base|base2) env="";;
    pipe2048) env="NRM_BT_PIPE=1";;
    pipe1024) env="NRM_BT_PIPE=1 NRM_BT_WAVES=1024";;
  esac
  env $env timeout -k 10 300 python bench.py --workload C3-large --steps 10 --warmup 3 --no-cpu-baseline --no-probe > gpurun_out/r5bt/$arm.json 2> gpurun_out/r5bt/$arm.err || { echo "$arm failed"; tail -5 gpurun_out/r5bt/$arm.err; exit 1; }
  python - $arm <<'PY'
import json,sys
d=json.loads(open(f"gpurun_out/r5bt/{sys.argv[1]}.json").read().strip().splitlines()[-1])
k=d["kernels"]
print(sys.argv[1], d["ms_per_step"], {n:round(k[n]["mean_ms"],3) if "mean_ms" in k[n] else k[n] for n in ("pwattn_bwd_e_bt","pwattn_bwd_e_bh","pwattn_bwd_e_dw","nrm_slab_reduce_multi")})
PY
done
