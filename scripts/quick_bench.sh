#!/bin/bash
# ON THE GPU BOX: one bench run, JSON kept under gpurun_out/quick/<tag>.json, one summary line on stdout.
#   usage: scripts/quick_bench.sh <tag> [bench.py arguments ...]
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-$PWD}
mkdir -p $R/gpurun_out/quick
timeout -k 10 300 python3 $R/bench.py --no-cpu-baseline "$@" > $R/gpurun_out/quick/$TAG.json 2> $R/gpurun_out/quick/$TAG.err || { echo "$TAG FAILED"; tail -5 $R/gpurun_out/quick/$TAG.err; exit 1; }
python3 - "$R/gpurun_out/quick/$TAG.json" "$TAG" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
k = {n: round(v["mean_ms"], 3) for n, v in d["kernels"].items() if "pwattn" in n and "pack" not in n}
print(sys.argv[2], d["value"], "imp/s", d["ms_per_step"], "ms", d["config"]["launch"], d["roofline"]["kernel"], d["roofline"]["frac"], k)
PY
