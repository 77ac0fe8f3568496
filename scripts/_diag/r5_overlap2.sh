#!/bin/bash
# step-level scheduling knobs re-measured with the dP walk + direct dW_p kernels (C3, eager): dW_p-last wait, attention streams, task counts
out=gpurun_out/r5ov; mkdir -p $out
run() { tag=$1; shift; env "$@" python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-probe --no-kernel-timing > $out/$tag.json 2> $out/$tag.err; python scripts/_diag/pr.py $out/$tag.json | sed "s|^|$tag: |"; }
for i in 1 2; do
  run base_$i X=0
  run dwlast0_$i NRM_DW_LAST=0
  run onestream_$i NRM_BRANCH_STREAMS=0
  run waves12k_$i NRM_BT_WAVES=12288
  run waves3k_$i NRM_BT_WAVES=3072
  run wgrad_$i NRM_WGRAD_STREAM=1
done
