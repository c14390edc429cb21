"""On-disk checkpoint formats of the reference, so real weights drop into the MI355X modules.

* DiT (hyvideo/inference.py:279-354, `Inference.load_state_dict`): `--dit-weight` is a file or a directory; a directory
  holds either `pytorch_model_<load_key>.pt` (bare state dict) or a deepspeed `*_model_states.pt` whose state dict sits under
  the key `load_key` ("module" or "ema"); a file of unknown kind is treated as wrapped when it has a "module"/"ema" entry.
  The state dict is loaded with strict=True: key names are the reference's module attribute names, which ours keep.
* FP8: `<dit_weight>.replace(".pt", "_map.pt")` maps each Linear's module path to its per-tensor scale
  (hyvideo/modules/fp8_optimization.py:85-100) - consumed by modules/fp8_optimization.convert_fp8_linear.
* VAE (hyvideo/vae/__init__.py:79-102): `<vae_path>/config.json` + `<vae_path>/pytorch_model.pt`, optionally wrapped in
  {"state_dict": ...} and/or with a "vae." key prefix.

Files are read ONLY with loaders that execute nothing from the file: torch.load(weights_only=True, mmap=True) for .pt and
safetensors for .safetensors (the reference calls torch.load with full unpickling, inference.py:338, vae/__init__.py:97).  A .pt
that the safe loader refuses (arbitrary pickled objects beside the tensors) raises CheckpointFormatError telling the user to
re-save the tensors; it is never unpickled."""
from __future__ import annotations

import json
import os
from pathlib import Path
from typing import Dict, Optional, Tuple

import torch


class CheckpointFormatError(ValueError):
    pass


def read_tensors(path) -> Dict:
    """Tensor container at `path` -> (possibly nested) dict, without executing anything from the file."""
    path = Path(path)
    if path.suffix == ".safetensors":
        from safetensors.torch import load_file
        return load_file(str(path), device="cpu")
    try:
        return torch.load(str(path), map_location="cpu", weights_only=True, mmap=True)
    except (RuntimeError, ValueError):       # not a zip archive (legacy format): mmap unsupported, same safe loader without it
        pass
    try:
        return torch.load(str(path), map_location="cpu", weights_only=True)
    except Exception as e:  # noqa: BLE001 - the safe unpickler's refusal; report, never fall back to full unpickling
        raise CheckpointFormatError(
            f"{path}: torch.load(weights_only=True) refused this file ({type(e).__name__}: {str(e).splitlines()[0][:200]}). "
            f"It holds pickled objects other than tensors; re-save its tensors (torch.save of the plain state dict, or safetensors) "
            f"in a trusted environment - this loader never unpickles arbitrary objects.") from e


def _pick_in_dir(d: Path, load_key: str) -> Tuple[Path, object]:
    files = sorted(d.glob("*.pt"))
    if len(files) == 0:
        raise ValueError(f"No model weights found in {d}")
    if files[0].name.startswith("pytorch_model_"):
        return d / f"pytorch_model_{load_key}.pt", True
    states = [f for f in files if f.name.endswith("_model_states.pt")]
    if states:
        return states[0], False
    raise ValueError(f"Invalid model path: {d} with unrecognized weight format: {list(map(str, files))}. When given a directory as "
                     f"--dit-weight, only `pytorch_model_*.pt` and `*_model_states.pt` (saved by deepspeed) can be parsed. If you "
                     f"want to load a specific weight file, please provide the full path to the file.")


def resolve_dit_weight(dit_weight, pretrained_model_path=None, model_resolution: str = "540p", load_key: str = "module"):
    """inference.py:281-336 -> (model_path, bare_model) with bare_model in {True, False, "unknown"}."""
    if dit_weight is None:
        if pretrained_model_path is None:
            raise ValueError("either dit_weight or pretrained_model_path is required")
        model_path, bare = _pick_in_dir(Path(pretrained_model_path) / f"t2v_{model_resolution}", load_key)
    else:
        dit_weight = Path(dit_weight)
        if dit_weight.is_dir():
            model_path, bare = _pick_in_dir(dit_weight, load_key)
        elif dit_weight.is_file():
            model_path, bare = dit_weight, "unknown"
        else:
            raise ValueError(f"Invalid model path: {dit_weight}")
    if not model_path.exists():
        raise ValueError(f"model_path not exists: {model_path}")
    return model_path, bare


def resolve_dit_path(args, pretrained_model_path=None) -> Path:
    """The checkpoint file load_state_dict(args, ...) will read (so `<file>_map.pt` can be located before loading)."""
    return resolve_dit_weight(getattr(args, "dit_weight", None), pretrained_model_path, getattr(args, "model_resolution", "540p"),
                              getattr(args, "load_key", "module"))[0]


def load_state_dict(args, model, pretrained_model_path=None):
    """Mirror of Inference.load_state_dict(args, model, pretrained_model_path) (inference.py:279-354); args carries
    .dit_weight, .load_key and .model_resolution."""
    load_key = getattr(args, "load_key", "module")
    model_path, bare = resolve_dit_weight(getattr(args, "dit_weight", None), pretrained_model_path,
                                          getattr(args, "model_resolution", "540p"), load_key)
    state_dict = read_tensors(model_path)
    if bare == "unknown" and ("ema" in state_dict or "module" in state_dict):
        bare = False
    if bare is False:
        if load_key not in state_dict:
            raise KeyError(f"Missing key: `{load_key}` in the checkpoint: {model_path}. The keys in the checkpoint are: "
                           f"{list(state_dict.keys())}.")
        state_dict = state_dict[load_key]
    model.load_state_dict(state_dict, strict=True)
    return model


def read_vae_checkpoint(vae_path) -> Tuple[Dict, Optional[Dict]]:
    """vae/__init__.py:86-102 -> (state dict with the "vae." prefix / "state_dict" wrapper removed, config.json or None)."""
    vae_path = Path(vae_path)
    ckpt_file = vae_path / "pytorch_model.pt"
    if not ckpt_file.exists():
        raise ValueError(f"VAE checkpoint not found: {ckpt_file}")
    ckpt = read_tensors(ckpt_file)
    if "state_dict" in ckpt:
        ckpt = ckpt["state_dict"]
    if any(k.startswith("vae.") for k in ckpt.keys()):
        ckpt = {k.replace("vae.", ""): v for k, v in ckpt.items() if k.startswith("vae.")}
    cfg = None
    cfg_file = vae_path / "config.json"
    if cfg_file.exists():
        with open(cfg_file, "r") as f:
            cfg = json.load(f)
    return ckpt, cfg


def fp8_map_path(dit_weight_path) -> str:
    """fp8_optimization.py:86."""
    return os.fspath(dit_weight_path).replace(".pt", "_map.pt")
