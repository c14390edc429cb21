#!/bin/bash
# One gpurun call: (1) the attention parity tests with ab_libs/<variant>.so swapped in for the product library, (2) a same-box
# A/B of the named libraries at the bench shape (tools/ab_attn.sh).   usage: check_and_ab.sh <variant> <baseline>
export HV_ALLOW_EXPERIMENT_LIB=1     # experiment libraries are swapped in below
set -o pipefail
V=$1; B=$2
cp hunyuanvideo_efficiency_amd/lib/libhv_kernels.so /tmp/libhv_kernels.product.so
cp ab_libs/$V.so hunyuanvideo_efficiency_amd/lib/libhv_kernels.so
timeout -k 10 900 python -m pytest -x -q tests/test_gpu_attention_v3.py tests/test_gpu_attention_split.py "tests/test_gpu_ops.py::test_attention" \
  tests/test_gpu_ops.py::test_attention_strided_fused_buffers tests/test_gpu_ops.py::test_attention_forced_rescale \
  tests/test_gpu_fullsize.py::test_attention_production_shape_sampled_rows tests/test_gpu_fullsize.py::test_attention_ulysses8_shape_kv_split_vs_single_pass \
  > gpurun_out/attn_${V}_tests.log 2>&1
rc=$?
tail -15 gpurun_out/attn_${V}_tests.log
cp /tmp/libhv_kernels.product.so hunyuanvideo_efficiency_amd/lib/libhv_kernels.so
if [ $rc -ne 0 ]; then echo "variant $V FAILED its parity tests (rc=$rc): no A/B"; exit 1; fi
bash tools/ab_attn.sh $B $V "${@:3}" 2>&1 | tee gpurun_out/attn_${B}_vs_${V}_ab.log
cp /tmp/libhv_kernels.product.so hunyuanvideo_efficiency_amd/lib/libhv_kernels.so
