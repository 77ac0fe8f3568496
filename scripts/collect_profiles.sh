#!/bin/bash
# Run ON THE GPU BOX (through gpurun): the default bench line, a rocprofv3 kernel-stats pass and three PMC passes of
# the same command (counters in their own passes, no API tracing beside them).  Output: gpurun_out/final/.
# Afterwards, in the container:  python scripts/pmc_summary.py   (copies the summaries into profiles/)
R=${GRAFT_REPO_ROOT:-$PWD}
mkdir -p $R/gpurun_out/final && cd /tmp && export TMPDIR=/tmp
timeout -k 10 500 python3 $R/bench.py --pcie > $R/gpurun_out/final/bench.json 2> $R/gpurun_out/final/bench.err; echo bench rc=$? &&
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/final/stats -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline > /dev/null 2>&1; echo stats rc=$? &&
timeout -k 10 200 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_LDS_BANK_CONFLICT --output-format csv -d $R/gpurun_out/final/sq -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2>&1; echo sq rc=$? &&
timeout -k 10 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE GRBM_GUI_ACTIVE --output-format csv -d $R/gpurun_out/final/fetch -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2>&1; echo fetch rc=$? &&
timeout -k 10 200 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/final/write -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2>&1; echo write rc=$?
