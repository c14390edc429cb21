import csv, collections, glob, sys
for d in sys.argv[1:]:
    f = glob.glob(f"{d}/*/*_counter_collection.csv")[0]
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if "attn_fwd" in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    g = {k: sum(v) / len(v) for k, v in agg.items()}
    wc = g["SQ_WAVE_CYCLES"] * 4
    print(f"{d}: wave_cycles={wc:.3e}  mfma_busy/wave_cycles*2={2*g['SQ_VALU_MFMA_BUSY_CYCLES']/wc:.3f}  wait_any={g['SQ_WAIT_ANY']*4/wc:.3f} wait_inst={g['SQ_WAIT_INST_ANY']*4/wc:.3f} active={g['SQ_ACTIVE_INST_ANY']*4/wc:.3f} active_valu={g['SQ_ACTIVE_INST_VALU']*4/wc:.3f} insts_valu={g['SQ_INSTS_VALU']:.3e}")
