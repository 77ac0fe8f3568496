mkdir -p gpurun_out/r5u
python -c "
import torch
print('priority range', torch.cuda.Stream.priority_range() if hasattr(torch.cuda.Stream,'priority_range') else None)
for p in (-2,-1,0,1,2):
    try:
        s=torch.cuda.Stream(priority=p); print(p, '->', s.priority)
    except Exception as e: print(p, 'ERR', e)
"
one() { tag=$1; shift; env "$@" > gpurun_out/r5u/$tag.json 2> gpurun_out/r5u/$tag.err; python -c "
import json; d=json.loads(open('gpurun_out/r5u/$tag.json').read().strip().splitlines()[-1]); print('$tag', d['ms_per_step'], d['config']['launch'], flush=True)"; }
for rep in 1 2 3; do
  one c3_prio_low_$rep NRM_BRANCH_PRIO=1 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --eager --no-probe
  one c3_prio_def_$rep A=1 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --eager --no-probe
  one c3_prio_high_$rep NRM_BRANCH_PRIO=-1 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --eager --no-probe
done
