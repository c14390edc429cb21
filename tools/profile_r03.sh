#!/bin/bash
# Round-3 counter evidence: per hot kernel, separate rocprofv3 --pmc passes (gpurun refuses --pmc together
# with trace domains other than --kernel-trace; FETCH_SIZE and WRITE_SIZE do not fit one pass).  Program directly after `--`.
# usage: tools/profile_r03.sh <target>...   (targets of tools/profile_r03_targets.py); results -> gpurun_out/prof_r03/<target>/
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/prof_r03; mkdir -p $O
SQ="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU GRBM_GUI_ACTIVE"
LDS="SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_VALU_MFMA_COEXEC_CYCLES"
for t in "$@"; do
  mkdir -p $O/$t
  i=0
  for set in "$SQ" "$LDS" "FETCH_SIZE" "WRITE_SIZE"; do
    i=$((i+1))
    timeout -k 10 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $O/$t/pass$i -- python3 tools/profile_r03_targets.py $t > $O/$t/pass$i.log 2>&1 \
      || { echo "$t pass $i failed"; tail -5 $O/$t/pass$i.log; exit 1; }
  done
  python3 tools/parse_pmc_r03.py $O/$t $t | tee $O/$t/summary.json
done
echo "profiles done"
