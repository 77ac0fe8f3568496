# eager stepping at C3 (what a multi-rank run does by default) with and without the side streams
mkdir -p gpurun_out/r5m
for cfg in "1 1" "0 0" "0 1" "1 0" "1 1" "0 0"; do
  set -- $cfg
  NRM_WGRAD_STREAM=$1 NRM_BRANCH_STREAMS=$2 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --eager > gpurun_out/r5m/e_w$1_b$2.json 2> gpurun_out/r5m/e_w$1_b$2.err
  python - <<PY
import json
d=json.loads(open("gpurun_out/r5m/e_w$1_b$2.json").read().strip().splitlines()[-1])
print("eager wgrad", $1, "branch", $2, d["ms_per_step"], flush=True)
PY
done
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --graph > gpurun_out/r5m/g.json 2>/dev/null; python -c "
import json; d=json.loads(open('gpurun_out/r5m/g.json').read().strip().splitlines()[-1]); print('graph default', d['ms_per_step'])"
