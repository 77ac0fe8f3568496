# the dW_p-only pass of the text+image attention behind the label attention's chain (NRM_DW_LAST): C3 eager, alternating, + C5, + graph
mkdir -p gpurun_out/r5q
one() { tag=$1; shift; env "$@" > gpurun_out/r5q/$tag.json 2> gpurun_out/r5q/$tag.err; python -c "
import json; d=json.loads(open('gpurun_out/r5q/$tag.json').read().strip().splitlines()[-1]); print('$tag', d['ms_per_step'], d['config']['launch'], flush=True)"; }
for rep in 1 2 3; do
  one c3_last1_$rep NRM_DW_LAST=1 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --eager
  one c3_last0_$rep NRM_DW_LAST=0 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --eager
done
one c3_last1_warm NRM_DW_LAST=1 python bench.py --steps 20 --warmup 40 --no-cpu-baseline --eager
one c3_last0_warm NRM_DW_LAST=0 python bench.py --steps 20 --warmup 40 --no-cpu-baseline --eager
one c3_last1_graph NRM_DW_LAST=1 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --graph
one c3_last0_graph NRM_DW_LAST=0 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --graph
one c5_last1 NRM_DW_LAST=1 python bench.py --workload C5-long --steps 10 --warmup 3 --no-cpu-baseline --eager
one c5_last0 NRM_DW_LAST=0 python bench.py --workload C5-long --steps 10 --warmup 3 --no-cpu-baseline --eager
one c2_last1 NRM_DW_LAST=1 python bench.py --workload C2-small --steps 30 --warmup 5 --no-cpu-baseline
one c2_last0 NRM_DW_LAST=0 python bench.py --workload C2-small --steps 30 --warmup 5 --no-cpu-baseline
