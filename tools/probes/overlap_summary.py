#!/usr/bin/env python3
"""Summarise a rocprofv3 --kernel-trace CSV: how much of the wall-clock between the first and the last kernel has 0 / 1 / 2+
kernels in flight, and which kernel classes spend their time overlapped with another kernel.
usage: overlap_summary.py <dir with *kernel_trace.csv> [skip_fraction]   -> JSON on stdout"""
import csv
import glob
import json
import sys
from collections import defaultdict

d = sys.argv[1]
skip = float(sys.argv[2]) if len(sys.argv) > 2 else 0.0
f = sorted(glob.glob(d + "/**/*kernel_trace.csv", recursive=True))[0]
rows = []
with open(f) as fh:
    for r in csv.DictReader(fh):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
t0, t1 = rows[0][0], max(r[1] for r in rows)
lo = t0 + int((t1 - t0) * skip)
rows = [r for r in rows if r[0] >= lo]
t0 = rows[0][0]


def cls(n):
    for k in ("gemm256x", "gemm_wreg", "gemm_rowln", "gemm256sq", "gemm256q", "attention_image", "attention_branch", "attention_mfma",
              "layernorm512", "layernorm", "gemm_skinny", "gemm_kernel", "splitk_reduce", "topk", "bridge", "combine", "cosine"):
        if k in n:
            return k
    return "other"


ev = []
for s, e, n in rows:
    ev.append((s, 1, cls(n)))
    ev.append((e, -1, cls(n)))
ev.sort()
active = defaultdict(int)
depth = 0
last = ev[0][0]
by_depth = defaultdict(int)
alone = defaultdict(int)
shared = defaultdict(int)
for t, d_, c in ev:
    dt = t - last
    if dt > 0:
        by_depth[min(depth, 3)] += dt
        for k, v in active.items():
            if v > 0:
                (alone if depth == 1 else shared)[k] += dt
    depth += d_
    active[c] += d_
    last = t
wall = ev[-1][0] - ev[0][0]
out = dict(file=f, kernels=len(rows), wall_ms=wall / 1e6, sum_durations_ms=sum(e - s for s, e, _ in rows) / 1e6,
           share_of_wall={str(k): round(v / wall, 4) for k, v in sorted(by_depth.items())},
           per_class_ms={k: dict(alone=round(alone[k] / 1e6, 1), with_another_kernel=round(shared[k] / 1e6, 1))
                         for k in sorted(set(alone) | set(shared), key=lambda k: -(alone[k] + shared[k]))})
print(json.dumps(out, indent=1))
