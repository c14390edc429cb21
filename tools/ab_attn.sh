#!/bin/bash
# A/B of prebuilt kernel libraries in ONE box (boxes differ by several % in sustained clocks): ab_libs/<name>.so, 2 rounds each
export HV_ALLOW_EXPERIMENT_LIB=1     # experiment libraries are swapped in below
for round in 1 2; do
  for v in "$@"; do
    cp ab_libs/$v.so hunyuanvideo_efficiency_amd/lib/libhv_kernels.so
    echo "== $v (round $round)"; python tools/bench_kernels.py attn 2>&1 | grep "S=119056\|S=32768"
  done
done
