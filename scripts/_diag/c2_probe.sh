for cfg in "base:" "wg1:NRM_WGRAD_STREAM=1" "dzslab:NRM_DZ_ROWS=0"; do
  tag=${cfg%%:*}; envs=${cfg#*:}
  env $envs python bench.py --workload C2-small --no-cpu-baseline --steps 30 > gpurun_out/r4_c2_$tag.json 2> gpurun_out/r4_c2_$tag.err
  python - <<PY
import json
d=json.loads(open("gpurun_out/r4_c2_$tag.json").read().strip().splitlines()[-1])
k=d["kernels"]
print("$tag", d["ms_per_step"], d["config"]["launch"], d["launch_probe"])
if "$tag"=="base":
    tot=0
    for n,v in sorted(k.items(), key=lambda kv:-kv[1]["launches"]*kv[1]["mean_ms"]):
        print(f"   {n:28s} {v['launches']//3:3d} x {v['mean_ms']*1e3:7.1f} us = {v['launches']*v['mean_ms']/3:.3f}"); tot+=v['launches']*v['mean_ms']/3
    print("   sum", tot, d["roofline"].get("step"))
PY
done
