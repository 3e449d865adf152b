#!/usr/bin/env python3
"""Developer tool: the STATIC mix of vector instructions of a kernel by issue-cost class, priced with the per-class costs that
tools/micro/valu_issue.hip measured at 4 waves per SIMD in true shader cycles (profiles/r03_valu_issue_microbench.txt).
The result is the issue cost per VALU instruction that the SIMD's vector port can at best sustain for THIS mix -- what the
guide's 2 cycles per wave64 instruction become for code made of VOP3 / packed / min-max / compare-select forms.
usage: tools/valu_mix.py [mangled kernel prefix ...]"""
import collections
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
csrc = os.path.join(ROOT, "rustray_amd", "csrc")
asm = os.path.join(ROOT, "build", "rr_api.s")
os.makedirs(os.path.dirname(asm), exist_ok=True)
if not os.path.exists(asm) or os.path.getmtime(asm) < max(os.path.getmtime(os.path.join(csrc, f)) for f in os.listdir(csrc)):
    subprocess.check_call(["hipcc", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math", "--offload-arch=gfx950", "-Wno-unused-function",
                           "-S", "--cuda-device-only", "-o", asm, "rr_api.hip"], cwd=csrc, stderr=subprocess.DEVNULL)
# cycles per instruction per SIMD at 4 waves per SIMD (s_memtime, slowest wave of the workgroup)
COST = {"vop2_arith": 2.55, "vop3_arith": 3.6, "packed_f32": 4.31, "minmax": 4.25, "cmp_select": 3.81, "transcendental": 8.0, "other": 3.6}


def classify(op):
    if op.startswith("v_pk_"):
        return "packed_f32"
    if re.match(r"v_(min|max|med)3?_", op):
        return "minmax"
    if op.startswith(("v_cmp", "v_cndmask")):
        return "cmp_select"
    if re.match(r"v_(rcp|rsq|sqrt|exp|log|sin|cos)_", op):
        return "transcendental"
    if op.endswith("_e32") or op.endswith("_sdwa") or op.endswith("_dpp"):
        return "vop2_arith"
    if re.match(r"v_(fma|mad|div_|mul_hi|mul_lo|lshl_add|add3|bfe|perm|readlane|writelane|lshl_or|and_or|or3|xad|add_lshl|bitop3|ldexp|cvt_pk|mov_b64|lshlrev_b64|lshrrev_b64|ashrrev_i64)", op) or op.endswith("_e64"):
        return "vop3_arith"
    return "other"


lines = open(asm).read().splitlines()
kernels = sys.argv[1:] or ["_Z15k_trace_closestILb1", "_Z14k_trace_shadowILb1", "_Z7k_shadeILb1"]
for prefix in kernels:
    start = [i for i, l in enumerate(lines) if l.startswith(prefix)][0]
    end = [i for i, l in enumerate(lines) if i > start and l.startswith(".Lfunc_end")][0]
    c = collections.Counter()
    for l in lines[start:end]:
        t = l.strip()
        if l.startswith("\t") and t.startswith("v_"):
            c[classify(t.split()[0])] += 1
    n = sum(c.values())
    avg = sum(COST[k] * v for k, v in c.items()) / n
    print(f"{prefix}: {n} static VALU instructions; mix " + ", ".join(f"{k} {100.0 * v / n:.0f} %" for k, v in c.most_common())
          + f"; priced at the measured per-class issue costs: {avg:.2f} cycles per instruction per SIMD")
