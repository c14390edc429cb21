#!/usr/bin/env python3
"""Summarise a rocprofv3 --kernel-trace CSV: per kernel (and per grid size for the attention kernel) count, mean,
min, max duration.  Usage: summarize_trace.py <kernel_trace.csv> [out.md]"""
import csv, sys, collections
rows = collections.defaultdict(list)
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        name = r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0]
        if "at::native" in r["Kernel_Name"] or name.startswith("void at::"):
            name = "torch (setup: synthetic weight/input generation, copies)"
        key = (name, int(r["Grid_Size_X"]) if ("attn_fwd" in name or "gemm" in name) else 0)
        rows[key].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
tot = sum(sum(v) for v in rows.values())
lines = ["| kernel | grid.x (threads) | launches | total ms | % | mean ms | min ms | max ms |", "|---|---|---|---|---|---|---|---|"]
for (name, g), v in sorted(rows.items(), key=lambda kv: -sum(kv[1])):
    if sum(v) / tot < 0.0005: continue
    lines.append(f"| {name} | {g or ''} | {len(v)} | {sum(v):.1f} | {100*sum(v)/tot:.2f} | {sum(v)/len(v):.3f} | {min(v):.3f} | {max(v):.3f} |")
out = "\n".join(lines)
print(out)
if len(sys.argv) > 2:
    open(sys.argv[2], "w").write(out + "\n")
