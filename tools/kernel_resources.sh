#!/bin/bash
# VGPRs / SGPRs / scratch / occupancy of every kernel of rt_kernels.hip as the compiler reports them (no GPU needed).
#   tools/kernel_resources.sh [extra hipcc flags] | column -t
cd "$(dirname "$0")/.."
/opt/rocm/bin/hipcc --offload-arch=gfx950 -std=c++17 -O3 -ffp-contract=off -fPIC -Iinclude -Iraytracer_amd/csrc --cuda-device-only -c raytracer_amd/csrc/rt_kernels.hip \
	-Rpass-analysis=kernel-resource-usage -o /dev/null "$@" 2>&1 | python3 -c '
import re, sys
cur = None; rows = {}
for line in sys.stdin:
    m = re.search(r"Function Name: (\S+)", line)
    if m: cur = m.group(1); rows[cur] = {}; continue
    m = re.search(r"remark:\s+(VGPRs|AGPRs|TotalSGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|LDS Size \[bytes/block\]|SGPRs Spill|VGPRs Spill): (\d+)", line)
    if m and cur: rows[cur][m.group(1).replace(" Spill", "Spill").split()[0]] = int(m.group(2))
import subprocess
for k, v in rows.items():
    name = subprocess.run(["c++filt", k], capture_output=True, text=True).stdout.strip()
    name = re.sub(r"\(anonymous namespace\)::", "", name).split("(")[0].replace("void ", "")
    print(name, "vgpr", v.get("VGPRs"), "agpr", v.get("AGPRs"), "sgpr", v.get("TotalSGPRs"), "sspill", v.get("SGPRsSpill"), "vspill", v.get("VGPRsSpill"), "scratch", v.get("ScratchSize"), "occ", v.get("Occupancy"), "lds", v.get("LDS"))
'
