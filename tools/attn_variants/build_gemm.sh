#!/bin/bash
# Build ab_libs/<name>.so = the product objects with hv_gemm.hip recompiled with extra flags (e.g. -DHV_GROUP_M=8) for tools/ab_gemm.sh.
# usage: tools/attn_variants/build_gemm.sh <name> [flags ...]
set -e
ROOT=$(cd "$(dirname "$0")/../.." && pwd)
C=$ROOT/hunyuanvideo_efficiency_amd/csrc; O=$ROOT/hunyuanvideo_efficiency_amd/lib/obj
name=$1; shift
mkdir -p $ROOT/ab_libs /tmp/gemm_$name
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -Wno-unused-result "$@" -c $C/hv_gemm.hip -o /tmp/gemm_$name/hv_gemm.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $ROOT/ab_libs/$name.so $O/hv_api.o $O/hv_attention.o $O/hv_attention_w4.o /tmp/gemm_$name/hv_gemm.o $O/hv_rowwise.o $O/hv_vae.o
