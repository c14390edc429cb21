#!/bin/bash
# Round-2 profile collection on the GPU box (run through gpurun): counters in their own passes (gpurun refuses --pmc together
# with the trace domains other than --kernel-trace), summaries copied to profiles/r02/ by the caller.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/prof_r02; mkdir -p $O
SQ="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU"
for v in 5 2; do
  export HV_ATTN_VER=$v
  timeout -k 10 300 rocprofv3 --pmc $SQ --kernel-trace --output-format csv -d $O/pmc_sq_v$v -- python3 tools/profile_attn.py > $O/pmc_sq_v$v.log 2>&1 || { echo "pmc sq v$v failed"; tail -5 $O/pmc_sq_v$v.log; exit 1; }
done
export HV_ATTN_VER=5
timeout -k 10 300 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_VALU_MFMA_COEXEC_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/pmc_lds_v5 -- python3 tools/profile_attn.py > $O/pmc_lds_v5.log 2>&1 || { echo "pmc lds failed"; tail -5 $O/pmc_lds_v5.log; }
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch_v5 -- python3 tools/profile_attn.py > $O/pmc_fetch_v5.log 2>&1 || { echo "pmc fetch failed"; exit 1; }
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write_v5 -- python3 tools/profile_attn.py > $O/pmc_write_v5.log 2>&1 || { echo "pmc write failed"; exit 1; }
python3 tools/parse_pmc.py $O/pmc_sq_v5 $O/pmc_sq_v2 | tee $O/attn_pmc_summary.txt
echo "profiles done"
