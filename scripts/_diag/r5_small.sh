# small-shape benches (graph replay) with the round-5 GEMM stage depth / pool split forced off and on
mkdir -p gpurun_out/r5e
for w in C1-demo ref-default C2-small; do
  for kc in 1 0; do
    if [ $kc = 1 ]; then export NRM_NT_KC=1 NRM_POOL_JSPLIT=0; else unset NRM_NT_KC NRM_POOL_JSPLIT; fi
    python bench.py --workload $w --steps 50 --warmup 5 --no-cpu-baseline > gpurun_out/r5e/${w}_old$kc.json 2> gpurun_out/r5e/${w}_old$kc.err
    python - <<PY
import json
d=json.loads(open("gpurun_out/r5e/${w}_old$kc.json").read().strip().splitlines()[-1])
k=d["kernels"]
print("$w", "round-4 forms" if $kc else "round-5 forms", d["ms_per_step"], "gemm_nt", k["nrm_gemm_nt"]["mean_ms"], "pool_bmm", k["nrm_pool_bmm"]["mean_ms"], flush=True)
PY
  done
done
unset NRM_NT_KC NRM_POOL_JSPLIT
