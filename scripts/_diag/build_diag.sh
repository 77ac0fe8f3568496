#!/bin/bash
# timing-only diagnostic builds of the library (WRONG results by construction): scripts/_diag/libnrm_<tag>.so
# use: NRM_HOTPATH_LIB=scripts/_diag/libnrm_noepi.so python bench.py --steps 5 --warmup 2 --no-cpu-baseline
cd "$(dirname "$0")/../.." || exit 1
C=news_recommendation_model_amd/csrc
for v in ${VARIANTS:-"noepi:-DNRM_DIAG_NOEPI=1" "noload:-DNRM_DIAG_NOLOAD=1" "neither:-DNRM_DIAG_NOEPI=1 -DNRM_DIAG_NOLOAD=1" "nox:-DNRM_DIAG_NOLOAD=2" "noy:-DNRM_DIAG_NOLOAD=3" "noatom:-DNRM_DIAG_NOATOM=1" "same:-DNRM_DIAG_NOLOAD=4" "fnoinit:-DNRM_DIAG_FWD=1" "fnodma:-DNRM_DIAG_FWD=2" "fnoz:-DNRM_DIAG_FWD=4" "fnogelu:-DNRM_DIAG_FWD=8" "fbare:-DNRM_DIAG_FWD=15"}; do
  tag=${v%%:*}; flags=${v#*:}
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -Iinclude $flags \
    $C/pwattn_fwd.hip $C/pwattn_fwd_rw.hip $C/pwattn_bwd.hip $C/gemm.hip $C/head.hip $C/pool_loss.hip $C/frontend.hip $C/capi.hip \
    -o scripts/_diag/libnrm_$tag.so || exit 1
done
ls -la scripts/_diag/*.so
