# C2 (bf16x3) with the wave-task count of the weight-gradient GEMM and the rows per workgroup of the resident-X GEMM varied
mkdir -p gpurun_out/r5i
run() { tag=$1; shift; env "$@" python bench.py --workload C2-small --steps 40 --warmup 5 --no-cpu-baseline > gpurun_out/r5i/$tag.json 2> gpurun_out/r5i/$tag.err
  python - <<PY
import json
d=json.loads(open("gpurun_out/r5i/$tag.json").read().strip().splitlines()[-1]); k=d["kernels"]
print("$tag", d["ms_per_step"], "gemm_nt", k["nrm_gemm_nt"]["mean_ms"], "gemm_tn", k["nrm_gemm_tn"]["mean_ms"], flush=True)
PY
}
run base A=1
run tn2048 NRM_TN_WAVES=2048
run tn4096 NRM_TN_WAVES=4096
run bm64 NRM_RX_BM=64
run bm128 NRM_RX_BM=128
run bm32 NRM_RX_BM=32
run base2 A=1
python -c "import __graft_entry__ as g; g.build(); g.smoke()"
