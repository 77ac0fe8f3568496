#!/bin/bash
# experiment: software-pipelined dh contraction (NRM_BH_PIPE=1, default) vs the serial-epilogue kernel (=0),
# each over a few wave-task counts (the pipelined kernel runs 2 waves/SIMD, the serial one 3)
for pipe in ${PIPES:-0 1}; do
for bh in ${BHS:-2048 4096 6144 12288}; do
  echo -n "PIPE=$pipe BH_WAVES=$bh  "; NRM_BH_PIPE=$pipe NRM_BH_WAVES=$bh timeout -k 10 200 python bench.py --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],2), {k:round(v['mean_ms'],3) for k,v in d['kernels'].items() if 'bwd_e' in k})"
done
done
