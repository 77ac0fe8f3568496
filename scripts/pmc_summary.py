#!/usr/bin/env python3
"""Compact what scripts/collect_profiles.sh wrote under gpurun_out/prof_<tag>/ into the committed files under profiles/:
kernel-stats CSV + summary, per-kernel PMC means, the HBM-traffic figures bench.py quotes, and the bench line.  Every
summary is stamped with the sha256 of the kernel sources it measured (build.sources_digest(); bench.py refuses to quote a
traffic figure whose stamp differs from the sources it runs) and with the commit they belong to.

usage: python scripts/pmc_summary.py <tag> [round]      files are named r<round>_<tag>_*      (round defaults to 4)"""
import collections
import csv
import glob
import json
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.chdir(ROOT)
sys.path.insert(0, ROOT)
from news_recommendation_model_amd import build          # noqa: E402

tag = sys.argv[1] if len(sys.argv) > 1 else "c3"
rnd = sys.argv[2] if len(sys.argv) > 2 else "4"
src = f"gpurun_out/prof_{tag}"
pre = f"profiles/r{rnd}_{tag}"
head = subprocess.run(["git", "rev-parse", "--short=12", "HEAD"], capture_output=True, text=True).stdout.strip()
dirty = bool(subprocess.run(["git", "status", "--porcelain", "--", "news_recommendation_model_amd/csrc", "include"],
                            capture_output=True, text=True).stdout.strip())
stamp = {"kernel_sources_sha256": build.sources_digest(), "git_head": head + ("+uncommitted kernel edits" if dirty else "")}

newest = lambda pat: max(glob.glob(pat), key=os.path.getmtime)      # gpurun merges every run into the same directory
stats = newest(f"{src}/stats/*/*_kernel_stats.csv")
trace = newest(f"{src}/stats/*/*_kernel_trace.csv")
shutil.copy(stats, f"{pre}_bench_kernel_stats.csv")
# steps in the profiled run = launches of the once-per-step Adam kernel
steps = sum(1 for r in csv.DictReader(open(trace)) if "adam_dev_kernel" in r["Kernel_Name"])
unit = "steps"
if steps == 0:      # --mode infer: no optimizer; one model forward packs W_p once per attention
    steps = sum(1 for r in csv.DictReader(open(trace)) if "pack_wp" in r["Kernel_Name"]) // 2
    unit = "model forwards (calls/step below = per forward)"
summary = subprocess.run([sys.executable, "scripts/rocprof_summary.py", f"{pre}_bench_kernel_stats.csv", str(max(steps, 1))],
                         capture_output=True, text=True, check=True).stdout
summary = f"# {json.dumps(stamp)}\n# rocprofv3 --kernel-trace --stats of: bench.py (see {pre}_bench.json 'config'), {steps} {unit} incl. warm-up\n" + summary
open(f"{pre}_bench_kernel_stats.summary.txt", "w").write(summary)
print(summary)

out = {}
for leg in ("sq", "fetch", "write"):
    path = newest(f"{src}/{leg}/*/*_counter_collection.csv")
    rows = [r for r in csv.DictReader(open(path)) if "nrm::" in r["Kernel_Name"]]
    name = lambda r: r["Kernel_Name"].split("(")[0].replace("void nrm::", "").replace("nrm::", "")   # noqa: E731
    # the same instantiation is also launched on small shapes (parity probe, side GEMMs): keep the largest grid only
    biggest = collections.defaultdict(int)
    for r in rows:
        biggest[name(r)] = max(biggest[name(r)], int(r["Grid_Size"]))
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in rows:
        if int(r["Grid_Size"]) == biggest[name(r)]:
            agg[name(r)][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, c in agg.items():
        out.setdefault(k, {}).update({n: sum(v) / len(v) for n, v in c.items()})
        out[k]["full_size_launches_" + leg] = len(next(iter(c.values())))
os.makedirs("profiles/pmc", exist_ok=True)
json.dump(dict(out, _stamp=stamp), open(f"profiles/pmc/r{rnd}_{tag}_pmc_counters.json", "w"), indent=1, sort_keys=True)


def pick(prefix):
    """the full-size instantiation of a kernel family = the one with the most wave cycles"""
    c = [k for k in out if k.startswith(prefix)]
    return max(c, key=lambda k: (out[k].get("SQ_WAVE_CYCLES", 0.0), out[k].get("FETCH_SIZE", 0.0))) if c else None


walk = pick("pwattn_fwd_walk_")            # the bf16 walk or (round 5) the fp32 walk of D = 64
fwd = walk if walk else pick("pwattn_fwd_rw_kernel") if (pick("pwattn_fwd_rw_kernel") and out[pick("pwattn_fwd_rw_kernel")].get("SQ_WAVE_CYCLES", 0) >
                                      out.get(pick("pwattn_fwd_kernel") or "", {}).get("SQ_WAVE_CYCLES", 0)) else pick("pwattn_fwd_kernel")
e_kernels = sorted((k for k in out if k.startswith("bwd_e_kernel")), key=lambda k: -out[k].get("SQ_WAVE_CYCLES", 0.0))
# bwd_e_kernel<KT, DT, KS, WITH_DW, EXACT, MMA, WITH_DT, XHL4>: the fp32 dW_p-only pass (text+image attention) is "<.., 0, false, false>"
dw = pick("bwd_dw_direct_kernel") or next((k for k in e_kernels if k.endswith(", 0, false, false>")), None)    # (round 5: the one-accumulator-set form)
bt = pick("bwd_dw_r32_kernel") or next((k for k in e_kernels if k != dw and (", true, true" in k or ", true, false" in k)), None)   # dW_p-only form (bf16), or the WITH_DW instantiation = the (b,t) pass
bh = pick("bwd_e_pipe_kernel") or next((k for k in e_kernels if k not in (bt, dw)), None)
names = {"nrm_pwattn_fwd": fwd, "nrm_pwattn_bwd_dz": pick("bwd_dz_"), "pwattn_bwd_e_bt": bt, "pwattn_bwd_e_dw": dw, "pwattn_bwd_e_bh": bh,
         "pwattn_bwd_rw_dtdh": pick("bwd_dp_rw_kernel"), "pwattn_bwd_dp_dtdh": pick("bwd_dp_walk_kernel")}
traffic = dict(stamp)
traffic["_note"] = ("rocprofv3 PMC (separate passes, scripts/collect_profiles.sh), mean per full-size launch. FETCH_SIZE/WRITE_SIZE are KB. "
                    "Per MI355X_MICROARCH.md FETCH_SIZE under-reports wide (16 B/lane) coalesced streaming reads by 2x on gfx950: "
                    "corrected_bytes doubles FETCH for these kernels, whose bulk reads are 16 B/lane (forward: LDS-DMA dwordx4 / "
                    "b128 operand loads; dz: float4 loads; the backward contractions: 16-byte operand loads); WRITE_SIZE is exact.")
for t, kn in names.items():
    if kn is None:
        continue
    g = out[kn]
    f, w = g.get("FETCH_SIZE", 0.0), g.get("WRITE_SIZE", 0.0)
    act = max(g.get("GRBM_GUI_ACTIVE", 0.0) / 8, 1.0)
    traffic[t] = {"kernel": kn, "FETCH_SIZE_KB": f, "WRITE_SIZE_KB": w, "corrected_bytes": (2 * f + w) * 1024,
                  "mfma_busy_frac": g.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / 1024 / act,
                  "clock_GHz_times_ms": g.get("GRBM_GUI_ACTIVE", 0.0) / 8 / 1e6,
                  "wait_any_frac": g.get("SQ_WAIT_ANY", 0.0) / max(g.get("SQ_WAVE_CYCLES", 1.0), 1.0),
                  "wait_inst_frac": g.get("SQ_WAIT_INST_ANY", 0.0) / max(g.get("SQ_WAVE_CYCLES", 1.0), 1.0),
                  "lds_bank_conflict_frac": g.get("SQ_LDS_BANK_CONFLICT", 0.0) / 256 / act}
    print(t, {k: (round(v, 4) if isinstance(v, float) else v) for k, v in traffic[t].items()})
json.dump(traffic, open(f"{pre}_pmc_traffic.json", "w"), indent=1)
shutil.copy(f"{src}/bench.json", f"{pre}_bench.json")
d = json.loads(open(f"{pre}_bench.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d["roofline"], d.get("cpu_baseline", {}).get("value"), d.get("pcie_inclusive"), d.get("fwd_auc_parity"))
