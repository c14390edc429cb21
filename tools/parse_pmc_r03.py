"""Fold the four rocprofv3 --pmc passes of tools/profile_r03.sh over one target into one JSON: mean counter values of the
target's dominant kernel per launch, the kernel's mean duration from the pass's own kernel trace, and the derived figures quoted
in DESIGN.md section 4 - matrix-pipe busy fraction, effective clock, memory-side bytes (FETCH_SIZE x 2: gfx950 counts 64 B per
128-B request for 16-B/lane streams, MI355X_MICROARCH.md section HBM; WRITE_SIZE exact) against the algorithmic bytes, GB/s.
SQ_* count quad-cycles except SQ_VALU_MFMA_BUSY_CYCLES (cycles)."""
import csv, glob, json, re, sys, collections
d, target = sys.argv[1], sys.argv[2]
algo = None
for lg in sorted(glob.glob(f"{d}/pass*.log")):
    for ln in open(lg, errors="replace"):
        if ln.startswith("ALGO "):
            algo = json.loads(ln[5:])
assert algo, "no ALGO line in the pass logs"
kname = algo["kernel"]
out = {"target": target, **algo}
counters, ms, waves_per_simd = {}, [], None
for p in sorted(glob.glob(f"{d}/pass*/")):
    cc = glob.glob(p + "*/*_counter_collection.csv")
    kt = glob.glob(p + "*/*_kernel_trace.csv")
    if not cc:
        continue
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(cc[0])):
        if kname in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in agg.items():
        counters[k] = sum(v) / len(v)
    if kt:
        t = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6 for r in csv.DictReader(open(kt[0])) if kname in r["Kernel_Name"]]
        if t:
            ms.append(sum(t) / len(t))
out["counters_mean_per_launch"] = counters
out["kernel_ms_per_pass"] = ms
t_s = (sum(ms) / len(ms)) * 1e-3
out["tflops"] = algo["flop"] / t_s / 1e12
if "SQ_WAVE_CYCLES" in counters:
    wc = counters["SQ_WAVE_CYCLES"] * 4
    # matrix pipe busy per SIMD: MFMA_BUSY is summed over waves; with w waves per SIMD the pipe's busy fraction is w * busy / wave_cycles.
    # every profiled kernel here runs 512-thread workgroups at one workgroup per CU = 2 waves per SIMD.
    waves_per_simd = 1 if "attn_fwd_kernel" in kname else 2      # attn_fwd_kernel_w4: 256-thread workgroups, one wave per SIMD
    out["mfma_busy_frac_per_simd"] = waves_per_simd * counters["SQ_VALU_MFMA_BUSY_CYCLES"] / wc
    out["wait_any_frac"] = counters["SQ_WAIT_ANY"] * 4 / wc
    out["wait_inst_any_frac"] = counters["SQ_WAIT_INST_ANY"] * 4 / wc
if "GRBM_GUI_ACTIVE" in counters and ms:
    out["effective_clock_ghz"] = counters["GRBM_GUI_ACTIVE"] / 8 / (ms[0] * 1e-3) / 1e9
if "FETCH_SIZE" in counters and "WRITE_SIZE" in counters:
    traffic = counters["FETCH_SIZE"] * 1024 * 2 + counters["WRITE_SIZE"] * 1024
    out["traffic_bytes_per_launch"] = traffic
    out["traffic_over_algorithmic"] = traffic / algo["bytes"]
    out["memory_side_GBps"] = traffic / t_s / 1e9
    out["algorithmic_GBps"] = algo["bytes"] / t_s / 1e9
print(json.dumps(out, indent=1))
