"""FP8 weight-only path with the reference's entry points (hyvideo/modules/fp8_optimization.py): convert_fp8_linear
marks every Linear under double_blocks / single_blocks, stores its weight as OCP float8_e4m3fn (the gfx950-native fp8
encoding - the same bits the reference stores) plus a per-tensor `fp8_scale`, and the forward dequantises per call
(ParamLinear.w() -> hv_fp8_dequant_bf16) before the bf16 MFMA GEMM: weight-only FP8, activations and accumulation
unchanged, ~12.5 GB of HBM instead of ~25 GB.

The offline quantiser (get_fp_maxval / quantize_to_fp8 / fp8_tensor_quant, :7-48) is restated with torch ops: it runs
once at conversion time, not in the denoise loop."""
from __future__ import annotations

import os
from typing import Dict, Optional

import torch

from .layers import ParamLinear

E4M3_MAX = 448.0   # get_fp_maxval(bits=8, mantissa_bit=3, sign_bits=1), fp8_optimization.py:7-18


def get_fp_maxval(bits: int = 8, mantissa_bit: int = 3, sign_bits: int = 1) -> float:
    m = max(1, min(mantissa_bit, bits - sign_bits))
    e = bits - sign_bits - m
    bias = 2 ** (e - 1) - 1
    mant = sum(1.0 / (2 ** i) for i in range(mantissa_bit))
    return mant * 2 ** (2 ** e - 1 - bias)


def quantize_to_fp8(x: torch.Tensor, bits: int = 8, mantissa_bit: int = 3, sign_bits: int = 1):
    """fp8_optimization.py:20-41: clamp to +-maxval, quantise the mantissa on the exponent grid; returns (qdq, step)."""
    m = max(1, min(mantissa_bit, bits - sign_bits))
    e = bits - sign_bits - m
    bias = 2 ** (e - 1) - 1
    maxval = get_fp_maxval(bits, mantissa_bit, sign_bits)
    xc = x.clamp(-maxval if sign_bits == 1 else 0.0, maxval)
    log_scales = torch.clamp(torch.floor(torch.log2(torch.abs(xc)) + bias), min=1.0)
    step = 2.0 ** (log_scales - m - bias)
    return torch.round(xc / step) * step, step


def fp8_tensor_quant(x: torch.Tensor, scale: torch.Tensor):
    """fp8_optimization.py:43-48."""
    s = scale.reshape([-1] + [1] * (x.dim() - 1)) if scale.dim() > 0 else scale
    q, step = quantize_to_fp8(x / s)
    return q, s, step


def quantize_weight(w: torch.Tensor):
    """The on-the-fly branch of fp8_linear_forward (:57-62): scale = max|W| / 448; W8 = e4m3fn(qdq(W / scale))."""
    scale = torch.max(torch.abs(w.flatten().float())) / get_fp_maxval()
    q, _, _ = fp8_tensor_quant(w.float(), scale)
    return q.to(torch.float8_e4m3fn), scale


def enable_fp8_mfma(module, enable: bool = True) -> int:
    """Opt-in FP8 COMPUTE (BASELINE.json configs[3] "fp8 weight path (CDNA4 fp8 MFMA)"): after convert_fp8_linear, every
    double / single block runs its Linears on v_mfma_scale_f32_16x16x128_f8f6f4 - the stored e4m3 weights go to the matrix cores
    as they are (no per-forward dequantisation, `fp8_scale` applied in the GEMM epilogue) and the activations are quantised per
    token to e4m3 (models._gemm / models._ln).  This is NOT the reference's arithmetic (its FP8 is weight-only): it adds the e4m3
    rounding of the activations; tolerance and the quantisation-aware oracle: tests/test_gpu_fp8_mfma.py."""
    n = 0
    for key, block in module.named_modules():
        if hasattr(block, "hybrid_seq_parallel_attn") and hasattr(block, "run"):
            block.fp8_mfma = bool(enable)
            n += 1
    return n


def convert_fp8_linear(module, dit_weight_path: Optional[str], original_dtype, params_to_keep={}, fp8_map: Optional[Dict] = None):
    """fp8_optimization.py:82-100.  `<dit_weight_path>_map.pt` holds the per-layer scales of a real FP8 checkpoint and is
    read with torch.load(weights_only=True) (never unpickled).  Without a checkpoint (random-init benchmarks) pass
    dit_weight_path=None: every layer is quantised from its current weights with scale = max|W|/448 - the reference's own
    fallback for non-fp8 weights (:57-62)."""
    setattr(module, "fp8_matmul_enabled", True)
    if fp8_map is None and dit_weight_path is not None:
        fp8_map_path = dit_weight_path.replace(".pt", "_map.pt")
        if not os.path.exists(fp8_map_path):
            raise ValueError(f"Invalid fp8_map path: {fp8_map_path}.")
        fp8_map = torch.load(fp8_map_path, map_location="cpu", weights_only=True)
    n = 0
    for key, layer in module.named_modules():
        if isinstance(layer, ParamLinear) and ("double_blocks" in key or "single_blocks" in key):
            w = layer.weight.data
            if fp8_map is not None:
                scale = fp8_map[key].to(device=w.device, dtype=torch.float32)
                w8 = w if w.dtype == torch.float8_e4m3fn else w.to(torch.float8_e4m3fn)
            else:
                w8, scale = quantize_weight(w)
            layer.weight = torch.nn.Parameter(w8.contiguous(), requires_grad=False)
            setattr(layer, "fp8_scale", scale.reshape(1).to(dtype=original_dtype).contiguous())
            n += 1
    return n
