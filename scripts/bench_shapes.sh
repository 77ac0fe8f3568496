#!/bin/bash
# the other BASELINE shapes (parity-test configs, not the bench line): 5 steps each, forward kernel time and step rate
for wl in ${WLS:-C2-small C1-demo C5-long ref-default}; do
  echo -n "$wl  "; timeout -k 10 300 python bench.py --workload $wl --steps 5 --warmup 2 --no-cpu-baseline ${EXTRA} 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], {k:round(v['mean_ms'],3) for k,v in d['kernels'].items() if 'pwattn' in k and 'pack' not in k})"
done
