# every committed profile of the round, final sources: bench line + rocprofv3 stats + three PMC passes per workload
bash scripts/collect_profiles.sh c3 &&
bash scripts/collect_profiles.sh c2_bf16x3 --workload C2-small &&
bash scripts/collect_profiles.sh refdefault --workload ref-default &&
bash scripts/collect_profiles.sh c1 --workload C1-demo &&
bash scripts/collect_profiles.sh infer_c3 --mode infer &&
bash scripts/collect_profiles.sh c5 --workload C5-long
