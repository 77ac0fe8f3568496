#!/bin/bash
# Diagnostic (round 5): gemm_nt LDS stage depth (NRM_NT_KC) at C3 -- the knob was only measured on the small shapes.
set -o pipefail
mkdir -p gpurun_out/r5kc
for arm in kc1 kc2 kc4 kc1b; do
  case $arm in kc1|kc1b) env="NRM_NT_KC=1";; kc2) env="NRM_NT_KC=2";; kc4) env="NRM_NT_KC=4";; esac
  env $env timeout -k 10 300 python bench.py --workload C3-large --steps 10 --warmup 3 --no-cpu-baseline --no-probe > gpurun_out/r5kc/$arm.json 2> gpurun_out/r5kc/$arm.err || { echo "$arm failed"; tail -5 gpurun_out/r5kc/$arm.err; exit 1; }
  python - $arm <<'PY'
import json,sys
d=json.loads(open(f"gpurun_out/r5kc/{sys.argv[1]}.json").read().strip().splitlines()[-1])
k=d["kernels"]
print(sys.argv[1], d["ms_per_step"], {n:(k[n]["launches"], round(k[n]["mean_ms"],4)) for n in ("nrm_gemm_nt","nrm_gemm_tn","nrm_pwattn_fwd")})
PY
done
