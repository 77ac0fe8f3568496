#!/bin/bash
for tag in ${TAGS:-noepi noload neither nox noy noatom same}; do
  echo -n "$tag  "; NRM_BH_PIPE=${PIPE:-1} NRM_HOTPATH_LIB=$PWD/scripts/_diag/libnrm_$tag.so timeout -k 10 200 python bench.py --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],2), {k:round(v['mean_ms'],3) for k,v in d['kernels'].items() if 'bwd_e' in k or k == 'nrm_pwattn_fwd'})"
done
