#!/bin/bash
# experiment: wave-task counts of the two backward contraction launches (env overrides read by capi.hip)
for bh in 2048 3072 4608 6144 9216 12288; do
  echo -n "BH_WAVES=$bh  "; NRM_BH_WAVES=$bh timeout -k 10 200 python bench.py --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print({k:round(v['mean_ms'],3) for k,v in d['kernels'].items() if 'bwd_e' in k})"
done
for bt in 1024 2048 4096 6144 8192; do
  echo -n "BT_WAVES=$bt  "; NRM_BT_WAVES=$bt timeout -k 10 200 python bench.py --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print({k:round(v['mean_ms'],3) for k,v in d['kernels'].items() if 'bwd_e' in k})"
done
