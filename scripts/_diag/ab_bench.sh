#!/bin/bash
# A/B of library builds on ONE box (box-to-box spread is larger than most kernel effects): product, each variant, product again.
#   usage: ab_bench.sh "<bench args>" variant1 [variant2 ...]      (variants: scripts/_diag/libnrm_<name>.so)
ARGS=$1; shift
for v in product "$@" product; do
  if [ $v = product ]; then unset NRM_HOTPATH_LIB; else export NRM_HOTPATH_LIB=scripts/_diag/libnrm_$v.so; fi
  python bench.py --no-cpu-baseline $ARGS > gpurun_out/ab_$v.json 2> gpurun_out/ab_$v.err
  python - <<PY
import json
d=json.loads(open("gpurun_out/ab_$v.json").read().strip().splitlines()[-1])
k=d["kernels"]
print("$v", d["ms_per_step"], d["config"]["launch"], {n:round(k[n]["mean_ms"],4) for n in k if k[n]["mean_ms"] > 0.15})
PY
done
