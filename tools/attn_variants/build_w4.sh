#!/bin/bash
# Build ab_libs/<name>.so = the product objects with the C ABI routed to the 4-wave x 64-row attention kernel (-DHV_ATTN_USE_W4=1);
# extra hipcc flags for hv_attention_w4.hip as further arguments (timing experiments).  Also prints the generated-code audit.
# usage: tools/attn_variants/build_w4.sh <name> [extra flags for hv_attention_w4.hip ...]
set -e
ROOT=$(cd "$(dirname "$0")/../.." && pwd)
C=$ROOT/hunyuanvideo_efficiency_amd/csrc; O=$ROOT/hunyuanvideo_efficiency_amd/lib/obj
name=$1; shift
make -C $C -j8 > /dev/null
mkdir -p $ROOT/ab_libs /tmp/w4_$name
F="--offload-arch=gfx950 -O3 -fPIC -std=c++17 -Wno-unused-result"
(cd /tmp/w4_$name && /opt/rocm/bin/hipcc $F "$@" -save-temps=obj -c $C/hv_attention_w4.hip -o /tmp/w4_$name/w4.o)
/opt/rocm/bin/hipcc $F -DHV_ATTN_USE_W4=1 -c $C/hv_attention.hip -o /tmp/w4_$name/attn.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $ROOT/ab_libs/$name.so /tmp/w4_$name/attn.o /tmp/w4_$name/w4.o $O/hv_api.o $O/hv_gemm.o $O/hv_rowwise.o $O/hv_vae.o
python3 $ROOT/tools/viz_w4_loop.py /tmp/w4_$name/hv_attention_w4-hip-amdgcn-amd-amdhsa-gfx950.s | head -14
