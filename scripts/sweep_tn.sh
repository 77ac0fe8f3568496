#!/bin/bash
for tn in 2048 4096 6144 12288; do
  echo -n "TN_WAVES=$tn  "; NRM_TN_WAVES=$tn timeout -k 10 200 python bench.py --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); k=d['kernels']; print(d['ms_per_step'], {n:(v['launches'],round(v['mean_ms'],3)) for n,v in k.items() if 'gemm' in n})"
done
