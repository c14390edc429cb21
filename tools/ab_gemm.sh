#!/bin/bash
# same-box A/B of prebuilt libraries on the GEMM shapes of the DiT step
export HV_ALLOW_EXPERIMENT_LIB=1
for round in 1 2; do
  for v in "$@"; do
    cp ab_libs/$v.so hunyuanvideo_efficiency_amd/lib/libhv_kernels.so
    echo "== $v (round $round)"; python tools/bench_kernels.py gemm 2>&1 | grep "gemm M"
  done
done
