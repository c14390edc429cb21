"""Summarise rocprofv3 --pmc passes over the attention kernel: per directory, the mean of every counter over the attention
dispatches and the ratios quoted in DESIGN.md (SQ_* count quad-cycles except SQ_VALU_MFMA_BUSY_CYCLES, which counts cycles:
MI355X_MICROARCH.md, cycle constants)."""
import csv, collections, glob, sys
for d in sys.argv[1:]:
    fs = glob.glob(f"{d}/*/*_counter_collection.csv")
    if not fs:
        print(f"{d}: no counter csv"); continue
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(fs[0])):
        if "attn_fwd" in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    g = {k: sum(v) / len(v) for k, v in agg.items()}
    line = f"{d}: " + "  ".join(f"{k}={v:.4e}" for k, v in sorted(g.items()))
    if "SQ_WAVE_CYCLES" in g:
        wc = g["SQ_WAVE_CYCLES"] * 4
        line += (f"\n    wave_cycles={wc:.3e}  matrix-pipe busy per SIMD (2 waves/SIMD) = 2*MFMA_BUSY/wave_cycles = {2*g['SQ_VALU_MFMA_BUSY_CYCLES']/wc:.3f}"
                 f"  wait_any={g['SQ_WAIT_ANY']*4/wc:.3f}  wait_inst={g['SQ_WAIT_INST_ANY']*4/wc:.3f}  active={g['SQ_ACTIVE_INST_ANY']*4/wc:.3f}"
                 f"  active_valu={g['SQ_ACTIVE_INST_VALU']*4/wc:.3f}  insts_valu={g['SQ_INSTS_VALU']:.3e}")
    print(line)
