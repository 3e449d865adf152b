#!/bin/bash
# Developer tool: register / spill / scratch / LDS table of the kernels of rr_api.hip for a set of -D flags (cross-compiles, no GPU).
# usage: tools/resusage.sh [-DRR_TRACE_WAVES=3 ...]
cd "$(dirname "$0")/../rustray_amd/csrc"
hipcc -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math --offload-arch=gfx950 -Wno-unused-function "$@" -Rpass-analysis=kernel-resource-usage -c rr_api.hip -o /dev/null 2>&1 | python3 -c "
import re, sys
cur = None; rows = {}
for line in sys.stdin:
    m = re.search(r'remark: +(Function Name|Name|TotalSGPRs|VGPRs|AGPRs|SGPRs Spill|VGPRs Spill|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|LDS Size \[bytes/block\]): (.*?) \[-Rpass', line)
    if not m:
        if 'error' in line: print(line.rstrip())
        continue
    k, v = m.group(1), m.group(2)
    if k in ('Function Name', 'Name'): cur = v; rows[cur] = {}
    elif cur: rows[cur][k] = v
import subprocess
for n, r in rows.items():
    try: d = subprocess.run(['c++filt', n], capture_output=True, text=True).stdout.split('(')[0].replace('void ', '')
    except Exception: d = n
    if not d.startswith('k_trace') and not d.startswith('k_shade') and '-a' not in sys.argv: continue
    print(f\"{d:28s} sgpr {r.get('TotalSGPRs','?'):>4s} vgpr {r.get('VGPRs','?'):>4s} agpr {r.get('AGPRs','?'):>3s} sspill {r.get('SGPRs Spill','?'):>4s} vspill {r.get('VGPRs Spill','?'):>3s} scratch {r.get('ScratchSize [bytes/lane]','?'):>4s} occ {r.get('Occupancy [waves/SIMD]','?'):>2s} lds {r.get('LDS Size [bytes/block]','?')}\")
"
