mkdir -p gpurun_out/quick
for v in 0 1; do
NRM_BRANCH_STREAMS=$v bash scripts/quick_bench.sh c2_bs$v --workload C2-small --steps 30 --warmup 5 | cut -c1-70
NRM_BRANCH_STREAMS=$v bash scripts/quick_bench.sh c1_bs$v --workload C1-demo --steps 30 --warmup 5 | cut -c1-70
NRM_BRANCH_STREAMS=$v bash scripts/quick_bench.sh rd_bs$v --workload ref-default --steps 30 --warmup 5 | cut -c1-70
NRM_BRANCH_STREAMS=$v bash scripts/quick_bench.sh c2e_bs$v --workload C2-small --steps 30 --warmup 5 --eager | cut -c1-70
NRM_BRANCH_STREAMS=$v bash scripts/quick_bench.sh c3_bs$v --steps 5 --warmup 2 | cut -c1-70
done
NRM_BRANCH_STREAMS=1 timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/quick/t_bs1.log 2>&1; echo rc=$?; tail -3 gpurun_out/quick/t_bs1.log
