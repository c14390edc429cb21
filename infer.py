#!/usr/bin/env python3
"""VAE reconstruction driver of the fork (infer.py:1-127) on the MI355X kernels: for every `<name>.pt` video tensor [C,T,H,W] in
--tensor-dir, run AutoencoderKLCausal3D.forward (encode -> posterior.mode() -> decode) under the temporal-op configuration of
--config-json (t_ops_config.json) and write the reconstruction to --output-dir/<name>.pt.  Same flags as the reference; input
tensors and checkpoints are read with torch.load(weights_only=True) only.  Without --vae-path the VAE gets deterministic
synthetic weights (there are no checkpoints in this environment): the plumbing and the kernels are what is exercised."""
import argparse
import os

import torch


class VideoTensorDataset:
    """dataset_processor/dataset_loader.py:9-24: sorted *.pt files, each a (C, T, H, W) tensor; returns (tensor, file name)."""

    def __init__(self, tensor_dir):
        self.tensor_dir = tensor_dir
        self.tensor_files = sorted(f for f in os.listdir(tensor_dir) if f.endswith(".pt"))

    def __len__(self):
        return len(self.tensor_files)

    def __getitem__(self, idx):
        path = os.path.join(self.tensor_dir, self.tensor_files[idx])
        return torch.load(path, map_location="cpu", weights_only=True), self.tensor_files[idx]


def infer_vae(model, dataset, device, output_dir, max_files=None):
    os.makedirs(output_dir, exist_ok=True)
    done = []
    for idx in range(len(dataset)):
        if max_files is not None and idx >= max_files:
            break
        video, file_name = dataset[idx]
        name = file_name.replace(".pt", "")
        video = video[None].to(device, dtype=torch.float16)              # the DataLoader's batch dimension (batch size 1)
        print(f"Processing {name}, video shape: {tuple(video.shape)}")
        with torch.no_grad():
            recon = model(video, return_dict=False, return_posterior=True, sample_posterior=False)[0]
        recon = recon.cpu().float()
        out_path = os.path.join(output_dir, f"{name}.pt")
        torch.save(recon, out_path)
        print(f"Saved reconstructed video to {out_path}, shape: {tuple(recon.shape)}")
        done.append(out_path)
    return done


def parse_args(argv=None):
    p = argparse.ArgumentParser(description="VAE inference script for video tensors (MI355X kernels).")
    p.add_argument("--tensor-dir", type=str, required=True, help="Directory containing input .pt video tensors.")
    p.add_argument("--output-dir", type=str, required=True, help="Directory to save the reconstructed videos.")
    p.add_argument("--vae-path", type=str, default=None, help="VAE checkpoint directory (config.json + pytorch_model.pt); "
                                                            "default: synthetic weights")
    p.add_argument("--config-json", type=str, default=None, help="Path to the T-ops config JSON file (t_ops_config.json).")
    p.add_argument("--max-files", type=int, default=None)
    p.add_argument("--mp4", action="store_true", help="accepted for flag compatibility; mp4 writing is outside this build (SURVEY 8f row 4)")
    p.add_argument("--batch-size", type=int, default=1)
    p.add_argument("--num-workers", type=int, default=4)
    p.add_argument("--reduced", action="store_true", help="synthetic-weight mode only: reduced channel widths (32,64,128,128)")
    return p.parse_args(argv)


def main(argv=None):
    a = parse_args(argv)
    if a.batch_size != 1:
        raise NotImplementedError("batch size 1 (the kernels process one video at a time)")
    device = "cuda"
    from hunyuanvideo_efficiency_amd import synthetic as syn
    from hunyuanvideo_efficiency_amd.vae import AutoencoderKLCausal3D, load_vae
    if a.vae_path:
        vae = load_vae("884-16c-hy", "fp16", vae_path=a.vae_path, device=device, t_ops_config_path=a.config_json, test=True,
                       with_encoder=True)[0]
    else:
        boc = (32, 64, 128, 128) if a.reduced else syn.VAE_BLOCK_OUT_CHANNELS
        vae = AutoencoderKLCausal3D(block_out_channels=boc, device=device, with_encoder=True)
        vae.load_state_dict({k: v.to(torch.float16) for k, v in syn.synth_vae_state_dict(boc, seed=0, encoder=True).items()}, strict=True)
        if a.config_json:
            from hunyuanvideo_efficiency_amd.vae import _apply_t_ops_config_to_vae, load_t_ops_config
            _apply_t_ops_config_to_vae(vae, load_t_ops_config(a.config_json))
    return infer_vae(vae, VideoTensorDataset(a.tensor_dir), device, a.output_dir, a.max_files)


if __name__ == "__main__":
    main()
