#!/bin/bash
# usage: tools/pmc_attn.sh <S> <variant...>  -> gpurun_out/pmc_v<variant>/
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
S=$1; shift
for v in "$@"; do
  timeout -k 10 200 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU --kernel-trace --output-format csv -d gpurun_out/pmc_v$v -- python tools/run_attn_once.py $S $v > gpurun_out/pmc_v$v.log 2>&1 || exit 1
done
