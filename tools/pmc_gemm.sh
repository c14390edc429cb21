#!/bin/bash
# FETCH_SIZE / WRITE_SIZE of the big fc1 GEMM (separate passes), kernel time alongside
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/prof_gemm; mkdir -p $O
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 200 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/pmc_$c -- python3 tools/profile_gemm.py > $O/pmc_$c.log 2>&1 || { echo "pmc $c failed"; tail -3 $O/pmc_$c.log; exit 1; }
done
python3 - <<'PY'
import csv, glob
O='gpurun_out/prof_gemm'
for c in ('FETCH_SIZE','WRITE_SIZE'):
    f=glob.glob(f"{O}/pmc_{c}/*/*_counter_collection.csv")[0]
    v=[float(r["Counter_Value"]) for r in csv.DictReader(open(f)) if "gemm8" in r["Kernel_Name"] and r["Counter_Name"]==c]
    print(c, "KB per launch:", sum(v)/len(v), "n", len(v))
f=glob.glob(f"{O}/pmc_FETCH_SIZE/*/*_kernel_trace.csv")[0]
ms=[(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e6 for r in csv.DictReader(open(f)) if "gemm8" in r["Kernel_Name"]]
print("kernel ms", ms)
PY
