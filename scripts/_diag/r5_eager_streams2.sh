# confirmation: C3 eager without / with the weight-gradient stream (two attention streams on), graph replay of both, C5 eager both
mkdir -p gpurun_out/r5n
one() { tag=$1; shift; env "$@" > gpurun_out/r5n/$tag.json 2> gpurun_out/r5n/$tag.err; python -c "
import json; d=json.loads(open('gpurun_out/r5n/$tag.json').read().strip().splitlines()[-1]); print('$tag', d['ms_per_step'], d['config']['launch'], flush=True)"; }
for rep in 1 2 3; do
  one c3_eager_w0_$rep NRM_WGRAD_STREAM=0 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --eager
  one c3_eager_w1_$rep NRM_WGRAD_STREAM=1 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --eager
done
one c3_graph_w0 NRM_WGRAD_STREAM=0 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --graph
one c3_graph_w1 NRM_WGRAD_STREAM=1 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --graph
one c3_probe_w0 NRM_WGRAD_STREAM=0 python bench.py --steps 20 --warmup 5 --no-cpu-baseline
one c5_eager_w0 NRM_WGRAD_STREAM=0 python bench.py --workload C5-long --steps 10 --warmup 3 --no-cpu-baseline --eager
one c5_eager_w1 NRM_WGRAD_STREAM=1 python bench.py --workload C5-long --steps 10 --warmup 3 --no-cpu-baseline --eager
