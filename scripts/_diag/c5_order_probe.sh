#!/bin/bash
# C5 dt+dW pass: HBM fetch (rocprofv3 --pmc FETCH_SIZE) and duration (--kernel-trace --stats) for both block orders of the (b,t)-grouped
# pass (NRM_BT_ORDER=0|1).  Run on the GPU box: bash scripts/_diag/c5_order_probe.sh
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/c5_order
mkdir -p $O && cd /tmp && export TMPDIR=/tmp
for ord in 0 1; do
  export NRM_BT_ORDER=$ord
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/fetch$ord -- python3 $R/scripts/_diag/attn_only.py 256 64 128 768 f32 2 bwd > $O/fetch$ord.log 2>&1; echo "fetch$ord rc=$?"
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats$ord -- python3 $R/scripts/_diag/attn_only.py 256 64 128 768 f32 4 bwd > $O/stats$ord.log 2>&1; echo "stats$ord rc=$?"
done
python3 - <<PY
import csv, glob, collections
for ord in (0, 1):
    f = max(glob.glob("$O/fetch%d/*/*_counter_collection.csv" % ord))
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if "bwd_e" in r["Kernel_Name"] and r["Counter_Name"] == "FETCH_SIZE":
            agg[r["Kernel_Name"].split("(")[0]].append(float(r["Counter_Value"]))
    for k, v in agg.items():
        print("order", ord, k, "FETCH_SIZE KB mean %.0f -> x2 correction %.2f GB per launch (%d launches)" % (sum(v) / len(v), 2 * sum(v) / len(v) * 1024 / 1e9, len(v)))
    f = max(glob.glob("$O/stats%d/*/*_kernel_stats.csv" % ord))
    for r in csv.DictReader(open(f)):
        if "bwd_e" in r["Name"]:
            print("order", ord, r["Name"].split("(")[0], "avg ms", float(r["AverageNs"]) / 1e6, "calls", r["Calls"])
PY
