#!/usr/bin/env python3
"""tools/kres.py FILE.hip [pattern] — per-kernel register / scratch / LDS usage as the compiler reports it"""
import re
import subprocess
import sys

src = sys.argv[1]
pat = sys.argv[2] if len(sys.argv) > 2 else ""
extra = sys.argv[3].split() if len(sys.argv) > 3 else []  # e.g. "-DEARHIP_BUILD_OCC=8"
cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-c", src, "-o",
       "/tmp/kres.o", "-Rpass-analysis=kernel-resource-usage"] + extra
out = subprocess.run(cmd, capture_output=True, text=True, cwd="/root/repo/libear_amd/csrc").stderr
cur, rows = None, {}
for ln in out.splitlines():
    m = re.search(r"remark:\s+(Function Name|VGPRs|AGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|SGPRs Spill|"
                  r"VGPRs Spill|LDS Size \[bytes/block\]|TotalSGPRs): (\S+)", ln)
    if not m:
        continue
    k, v = m.groups()
    if k == "Function Name":
        cur = v
        rows[cur] = {}
    elif cur:
        rows[cur][k.split(" [")[0]] = v
for name, r in rows.items():
    d = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip().split("(")[0]
    if pat and pat not in d:
        continue
    g = r.get
    print(f"{d:58s} VGPR {g('VGPRs'):>4} SGPR {g('TotalSGPRs'):>4} scratch {g('ScratchSize'):>4} occ {g('Occupancy')} "
          f"vspill {g('VGPRs Spill')} sspill {g('SGPRs Spill')} lds {g('LDS Size')}")
