#!/bin/bash
# same-box A/B: ab_run.sh "<command>" <lib name>...   (prebuilt ab_libs/<name>.so swapped in turn, two rounds)
cmd="$1"; shift
for round in 1 2; do
  for v in "$@"; do
    cp ab_libs/$v.so hunyuanvideo_efficiency_amd/lib/libhv_kernels.so
    echo "== $v (round $round)"; bash -c "$cmd" 2>&1 | grep -v amdgpu.ids
  done
done
