#!/bin/bash
# dP walk backward (NRM_BWD_DP=1) against the E-form at C3, same box, alternating; parity tests first
set -o pipefail
out=gpurun_out/r5dp; mkdir -p $out
python -m pytest tests/test_gpu_attention.py -x -q -k "dp_walk" > $out/tests.log 2>&1; echo "tests rc=$?" | tee -a $out/tests.log
tail -5 $out/tests.log
grep -q "rc=0" $out/tests.log || exit 1
for i in 1 2; do
  for m in 0 1; do
    NRM_BWD_DP=$m python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-probe > $out/c3_dp${m}_$i.json 2> $out/c3_dp${m}_$i.err || { echo "bench failed dp=$m"; tail -5 $out/c3_dp${m}_$i.err; exit 1; }
    python - <<P
import json
d=json.loads(open("$out/c3_dp${m}_$i.json").read().strip().splitlines()[-1])
k=d.get("kernels",{})
print("dp=$m run $i ms_per_step", d["ms_per_step"], {n:round(v["mean_ms"],3) for n,v in k.items() if "pwattn" in n})
P
  done
done
