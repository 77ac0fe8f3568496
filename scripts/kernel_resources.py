#!/usr/bin/env python3
"""VGPR / scratch / occupancy per kernel of one csrc/*.hip file (hipcc -Rpass-analysis=kernel-resource-usage), one line each.
    python scripts/kernel_resources.py pwattn_bwd_rw.hip [extra hipcc flags]"""
import os, re, subprocess, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(root, "news_recommendation_model_amd", "csrc", sys.argv[1])
cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-c", src, "-o", "/dev/null",
       "-Rpass-analysis=kernel-resource-usage", *sys.argv[2:]]
out = subprocess.run(cmd, capture_output=True, text=True).stderr
cur = None
rows = {}
for line in out.splitlines():
    m = re.search(r"remark: [^ ]+ +(Function )?Name: (\S+)", line) or re.search(r"Function Name: (\S+)", line)
    if "Name:" in line:
        cur = line.split("Name:")[1].split()[0]
        rows[cur] = {}
        continue
    m = re.search(r"(VGPRs|AGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|SGPRs|VGPRs Spill|SGPRs Spill|LDS Size \[bytes/block\]): (\d+)", line)
    if m and cur:
        rows[cur][m.group(1)] = int(m.group(2))
for k, v in rows.items():
    name = subprocess.run(["c++filt", k], capture_output=True, text=True).stdout.strip()[:110]
    print(f"{name:110s} vgpr {v.get('VGPRs','?'):>4} agpr {v.get('AGPRs','?'):>3} scratch {v.get('ScratchSize [bytes/lane]','?'):>4} "
          f"occ {v.get('Occupancy [waves/SIMD]','?')} sgpr {v.get('SGPRs','?')} lds {v.get('LDS Size [bytes/block]','?')}")
if not rows:
    print(out[-3000:])
