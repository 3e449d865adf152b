#!/usr/bin/env python3
"""Developer tool: STATIC vector-instruction count per source line of one kernel (hipcc -S -gline-tables-only, .loc directives).
Straight-line shading code executes once per hit, so for k_shade the static count is close to the dynamic one per path.
usage: tools/static_cost.py <mangled kernel prefix, e.g. _Z7k_shadeILb1> [top N]"""
import collections
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
asm = os.path.join(ROOT, "build", "rr_api_g.s")
csrc = os.path.join(ROOT, "rustray_amd", "csrc")
if not os.path.exists(asm) or os.path.getmtime(asm) < max(os.path.getmtime(os.path.join(csrc, f)) for f in os.listdir(csrc)):
    subprocess.check_call(["hipcc", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math", "--offload-arch=gfx950", "-Wno-unused-function",
                           "-gline-tables-only", "-S", "--cuda-device-only", "-o", asm, "rr_api.hip"], cwd=csrc, stderr=subprocess.DEVNULL)
lines = open(asm).read().splitlines()
files = {}
for l in lines:
    m = re.match(r'\s*\.file\s+(\d+)\s+"([^"]*)"(?:\s+"([^"]*)")?', l)
    if m:
        files[int(m.group(1))] = (m.group(3) or m.group(2)).split("/")[-1]
prefix = sys.argv[1]
top = int(sys.argv[2]) if len(sys.argv) > 2 else 50
start = [i for i, l in enumerate(lines) if l.startswith(prefix)][0]
end = [i for i, l in enumerate(lines) if i > start and l.startswith(".Lfunc_end")][0]
cur, cnt = None, collections.Counter()
for l in lines[start:end]:
    m = re.match(r"\s*\.loc\s+(\d+)\s+(\d+)", l)
    if m:
        cur = (files.get(int(m.group(1)), "?"), int(m.group(2)))
        continue
    t = l.strip()
    if l.startswith("\t") and t and not t.startswith((".", ";")) and t.split()[0].startswith("v_"):
        cnt[cur] += 1
print(prefix, "static VALU instructions:", sum(cnt.values()))
byfile = collections.Counter()
for (f, _), c in cnt.items():
    byfile[f] += c
print(byfile.most_common(6))
src = {f: open(os.path.join(csrc, f)).read().splitlines() for f in ("rr_kernels.hip", "rr_math.h")}
for f in src:
    items = sorted(((l, c) for (ff, l), c in cnt.items() if ff == f), key=lambda x: -x[1])[:top]
    for l, c in sorted(items):
        print(f"{f}:{l:5d} {c:5d}  {src[f][l - 1].strip()[:120]}")
