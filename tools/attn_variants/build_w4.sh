#!/bin/bash
# Build ab_libs/<name>.so = the product objects + an EXPERIMENT build of the 4-wave x 64-row attention kernel.  The steady-state iteration is
# generated HERE from the HV_W4_* variables of the calling environment (tools/gen_attn_w4_asm.py) into /tmp and included through
# -DHV_W4_LOOP_INC: the in-tree .inc and the product library are never touched (an experiment that regenerates the in-tree file and runs
# `make` leaves an experiment in hunyuanvideo_efficiency_amd/lib - that happened once).  Extra hipcc flags for hv_attention_w4.hip as further
# arguments; -DHV_W4_STAMPS also switches the generator to the stamped iteration.  Prints the generated-code audit.
# usage: [HV_W4_ORDER=.. HV_W4_ABL=.. ...] tools/attn_variants/build_w4.sh <name> [extra flags ...]
set -e
ROOT=$(cd "$(dirname "$0")/../.." && pwd)
C=$ROOT/hunyuanvideo_efficiency_amd/csrc; O=$ROOT/hunyuanvideo_efficiency_amd/lib/obj
name=$1; shift
mkdir -p $ROOT/ab_libs /tmp/w4_$name
case " $* " in *" -DHV_W4_STAMPS "*) export HV_W4_STAMPS=1;; esac
HV_W4_INC_OUT=/tmp/w4_$name/loop.inc python3 $ROOT/tools/gen_attn_w4_asm.py > /dev/null
F="--offload-arch=gfx950 -O3 -fPIC -std=c++17 -Wno-unused-result"
(cd /tmp/w4_$name && /opt/rocm/bin/hipcc $F "$@" -DHV_W4_LOOP_INC="\"/tmp/w4_$name/loop.inc\"" -save-temps=obj -c $C/hv_attention_w4.hip -o /tmp/w4_$name/w4.o)
/opt/rocm/bin/hipcc $F -c $C/hv_attention.hip -o /tmp/w4_$name/attn.o
for f in hv_api hv_gemm hv_rowwise hv_vae; do [ -f $O/$f.o ] || { echo "build the product first (make -C $C)"; exit 1; }; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $ROOT/ab_libs/$name.so /tmp/w4_$name/attn.o /tmp/w4_$name/w4.o $O/hv_api.o $O/hv_gemm.o $O/hv_rowwise.o $O/hv_vae.o
python3 $ROOT/tools/viz_w4_loop.py /tmp/w4_$name/hv_attention_w4-hip-amdgcn-amd-amdhsa-gfx950.s | head -14
