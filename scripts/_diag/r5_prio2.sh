mkdir -p gpurun_out/r5v
one() { tag=$1; shift; env "$@" > gpurun_out/r5v/$tag.json 2> gpurun_out/r5v/$tag.err; python -c "
import json; d=json.loads(open('gpurun_out/r5v/$tag.json').read().strip().splitlines()[-1]); print('$tag', d['ms_per_step'], d['config']['launch'], flush=True)"; }
for rep in 1 2 3; do
  one c3_mainhigh_$rep NRM_MAIN_PRIO=-1 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --eager --no-probe
  one c3_def_$rep A=1 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --eager --no-probe
done
