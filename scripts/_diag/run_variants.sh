#!/bin/bash
# usage: run_variants.sh "<tags>" <B> <T> <H> <D> <mma>   -- attn_kernels.py against the product library and each variant build
tags=$1; shift
python scripts/_diag/attn_kernels.py "$@" 2>/dev/null
for tag in $tags; do
  NRM_ALLOW_DIAG_LIB=1 NRM_HOTPATH_LIB=$PWD/scripts/_diag/libnrm_$tag.so timeout -k 10 120 python scripts/_diag/attn_kernels.py "$@" 2>/dev/null || echo "$tag FAILED"
done
