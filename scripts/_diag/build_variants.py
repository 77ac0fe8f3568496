#!/usr/bin/env python3
"""Variant builds of the library for A/B timing on the GPU box: scripts/_diag/libnrm_<tag>.so, one per "tag:flags" argument
(tuning macros give correct results; NRM_DIAG_* macros give timing-only builds that native.load refuses unless
NRM_ALLOW_DIAG_LIB=1).    usage: build_variants.py "la4:-DBRW_LA=4" "w8:-DBRW_WAVES_N=8 -DBRW_LA=6" ..."""
import os
import sys

root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, root)
from news_recommendation_model_amd import build

for spec in sys.argv[1:]:
    tag, flags = spec.split(":", 1)
    out = os.path.join(root, "scripts", "_diag", f"libnrm_{tag}.so")
    build.build(force=False, verbose=False, extra_flags=tuple(flags.split()), lib=out)
    print(out)
