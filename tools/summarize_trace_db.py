#!/usr/bin/env python3
"""Summarise a rocprofv3 --kernel-trace result database (rocpd sqlite, the default output of ROCm 7.2's rocprofv3): per kernel
(and per grid size for the attention / GEMM kernels) launches, total, share, mean / min / max duration, VGPRs, LDS.
Usage: summarize_trace_db.py <results.db> [out.md] [out.csv]"""
import collections, csv, sqlite3, sys
c = sqlite3.connect(sys.argv[1])
rows = collections.defaultdict(list)
meta = {}
for name, gx, gy, dur, vgpr, lds in c.execute("select name, grid_x, grid_y, duration, vgpr_count, lds_size from kernels"):
    short = name.replace("(anonymous namespace)::", "").split("(")[0]
    if "at::native" in name or short.startswith("void at::") or "rocprim" in name or "elementwise" in name:
        short = "torch (setup: synthetic weight/input generation, copies)"
    key = (short, (gx, gy) if ("attn_" in short or "gemm" in short) else (0, 0))
    rows[key].append(dur / 1e6)
    meta[key] = (vgpr, lds)
tot = sum(sum(v) for v in rows.values())
hdr = ["kernel", "grid (threads x,y)", "launches", "total_ms", "percent", "mean_ms", "min_ms", "max_ms", "vgpr", "lds_bytes"]
table = []
for key, v in sorted(rows.items(), key=lambda kv: -sum(kv[1])):
    (name, g) = key
    table.append([name, f"{g[0]}x{g[1]}" if g[0] else "", len(v), f"{sum(v):.1f}", f"{100 * sum(v) / tot:.2f}", f"{sum(v) / len(v):.3f}",
                  f"{min(v):.3f}", f"{max(v):.3f}", meta[key][0], meta[key][1]])
md = ["| " + " | ".join(hdr) + " |", "|" + "---|" * len(hdr)] + ["| " + " | ".join(map(str, r)) + " |" for r in table if float(r[4]) >= 0.05]
print("\n".join(md))
if len(sys.argv) > 2:
    open(sys.argv[2], "w").write("\n".join(md) + "\n")
if len(sys.argv) > 3:
    with open(sys.argv[3], "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(hdr)
        w.writerows(table)
