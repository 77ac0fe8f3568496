#!/bin/bash
# Run ON THE GPU BOX (through gpurun): the bench line, a rocprofv3 kernel-stats pass and three PMC passes of the SAME
# command (counters in their own passes, no API tracing beside them; the program itself follows "--").
#   usage: scripts/collect_profiles.sh <tag> [bench.py arguments ...]        e.g.  c3   |   c2_bf16x3 --workload C2-small
# Output: gpurun_out/prof_<tag>/.  Afterwards, in the container:  python scripts/pmc_summary.py <tag> <round>
TAG=${1:-c3}; shift
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/prof_$TAG
mkdir -p $O && cd /tmp && export TMPDIR=/tmp
timeout -k 10 500 python3 $R/bench.py "$@" > $O/bench.json 2> $O/bench.err; echo bench rc=$?
# the profiled passes run every kernel alone on the chip (one attention stream, no weight-gradient stream), as the steps bench.py
# takes its per-kernel durations from do: durations of overlapped kernels cannot be priced against a roofline
export NRM_BRANCH_STREAMS=0 NRM_WGRAD_STREAM=0
true &&
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 $R/bench.py "$@" --steps 5 --warmup 2 --no-cpu-baseline --eager > /dev/null 2>&1; echo stats rc=$? &&
timeout -k 10 200 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_LDS_BANK_CONFLICT --output-format csv -d $O/sq -- python3 $R/bench.py "$@" --steps 2 --warmup 1 --no-cpu-baseline --eager > /dev/null 2>&1; echo sq rc=$? &&
timeout -k 10 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE GRBM_GUI_ACTIVE --output-format csv -d $O/fetch -- python3 $R/bench.py "$@" --steps 2 --warmup 1 --no-cpu-baseline --eager > /dev/null 2>&1; echo fetch rc=$? &&
timeout -k 10 200 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/write -- python3 $R/bench.py "$@" --steps 2 --warmup 1 --no-cpu-baseline --eager > /dev/null 2>&1; echo write rc=$?
