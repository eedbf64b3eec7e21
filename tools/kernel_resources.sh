#!/bin/bash
# VGPRs / scratch (spills) / occupancy of every kernel in one translation unit:  tools/kernel_resources.sh ladder_toric.hip
cd "$(dirname "$0")/../mcmc-qec-toric-rl_amd/csrc"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -c -o /dev/null "$1" -Rpass-analysis=kernel-resource-usage 2>&1 | python3 -c '
import re, sys
name = None; row = {}
for line in sys.stdin:
    m = re.search(r"Function Name: (\S+)", line)
    if m:
        name = m.group(1); row = {}; continue
    for key in ("VGPRs", "ScratchSize \[bytes/lane\]", "Occupancy \[waves/SIMD\]", "LDS Size \[bytes/block\]"):
        m = re.search(r"\s" + key + r": (\d+)", line)
        if m: row[key.split()[0]] = int(m.group(1))
    if "LDS Size" in line and name:
        t = re.search(r"ILi(\d+)ELi(\d+)E((?:Lb\dE|Li\dE)+)", name)
        flags = re.findall(r"L[bi](\d)E", t.group(3)) if t else []
        print((t.group(1), t.group(2), "".join(flags)) if t else name[:60], row)
        name = None
'
