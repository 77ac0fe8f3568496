# the same comparison on a chip that has been under load as long as bench.py's default run keeps it (probe + capture) before its timed region
mkdir -p gpurun_out/r5o
one() { tag=$1; shift; env "$@" > gpurun_out/r5o/$tag.json 2> gpurun_out/r5o/$tag.err; python -c "
import json; d=json.loads(open('gpurun_out/r5o/$tag.json').read().strip().splitlines()[-1]); print('$tag', d['ms_per_step'], d['config']['launch'], d['launch_probe']['t_eager_ms'], d['launch_probe']['t_graph_ms'], flush=True)"; }
one eager_w0_warm40 NRM_WGRAD_STREAM=0 python bench.py --steps 20 --warmup 40 --no-cpu-baseline --eager
one eager_w1_warm40 NRM_WGRAD_STREAM=1 python bench.py --steps 20 --warmup 40 --no-cpu-baseline --eager
one graph_w1_warm20 NRM_WGRAD_STREAM=1 python bench.py --steps 20 --warmup 20 --no-cpu-baseline --graph
one graph_w0_warm20 NRM_WGRAD_STREAM=0 python bench.py --steps 20 --warmup 20 --no-cpu-baseline --graph
one eager_w0_steps100 NRM_WGRAD_STREAM=0 python bench.py --steps 100 --warmup 5 --no-cpu-baseline --eager
one probe_w0 NRM_WGRAD_STREAM=0 python bench.py --steps 20 --warmup 5 --no-cpu-baseline
one probe_w1 NRM_WGRAD_STREAM=1 python bench.py --steps 20 --warmup 5 --no-cpu-baseline
