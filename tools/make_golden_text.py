#!/usr/bin/env python3
"""Generate tests/golden/text_encoder.npz by IMPORTING the reference's text-encoder wrapper (/root/reference/hyvideo/text_encoder,
importable in the build container: it needs only torch + transformers) and running its `TextEncoder.text2tokens` / `encode`
(:220-339) on tiny random LLaMA / CLIP models.  No checkpoint or tokenizer files exist in this environment, so the script writes
its own: a 4-layer LlamaModel and a 2-layer CLIPTextModel (random weights from a seeded generator) with toy tokenizers, saved
with `save_pretrained` into a temporary directory that the REFERENCE's own `load_text_encoder` / `load_tokenizer`
(`from_pretrained`, :17-74) then reads.  Frozen: token ids, attention masks and hidden states for the video and image templates
(`hidden_state_skip_layer=2`, `apply_final_norm=True`, cropping), the plain last-layer path, and CLIP's pooled output - plus every
weight of the two tiny models (a few hundred KB), so the test rebuilds the same models without a generator-version dependence.
The reference never travels to the GPU box; only these vectors do.   Run:  python tools/make_golden_text.py"""
import json
import os
import sys
import tempfile

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "tests", "golden", "text_encoder.npz")
sys.dont_write_bytecode = True
sys.path.insert(0, "/root/reference")

sys.path.insert(0, ROOT)
from tests.toy_text_models import (WORDS, TPL_VIDEO, TPL_IMAGE, LLM_CFG, CLIP_CFG, toy_llm_tokenizer,  # noqa: E402,F401
                                   write_clip_tokenizer_files)


def main():
    from transformers import CLIPTextConfig, CLIPTextModel, LlamaConfig, LlamaModel
    import hyvideo.text_encoder as REF                      # the reference, executed here
    out = {}
    with tempfile.TemporaryDirectory() as tmp:
        # ---------------- LLM
        llm_dir = os.path.join(tmp, "llm")
        torch.manual_seed(0)
        llm = LlamaModel(LlamaConfig(**LLM_CFG)).eval()
        llm.save_pretrained(llm_dir)
        toy_llm_tokenizer().save_pretrained(llm_dir)
        for k, v in llm.state_dict().items():
            out["llm.w." + k] = v.detach().float().numpy()
        enc = REF.TextEncoder("llm", max_length=8 + TPL_VIDEO["crop_start"], text_encoder_precision="fp32", text_encoder_path=llm_dir,
                              tokenizer_type="llm", prompt_template=TPL_IMAGE, prompt_template_video=TPL_VIDEO, hidden_state_skip_layer=2,
                              apply_final_norm=True)
        for name, text, dt in (("video", "a cat walks on grass", "video"), ("image", "red car", "image"),
                               ("video_long", "a cat walks on grass slowly a cat walks on grass slowly", "video")):
            toks = enc.text2tokens(text, data_type=dt)
            o = enc.encode(toks, data_type=dt)
            out[f"llm.{name}.input_ids"] = toks["input_ids"].numpy()
            out[f"llm.{name}.attention_mask"] = toks["attention_mask"].numpy()
            out[f"llm.{name}.hidden_state"] = o.hidden_state.float().numpy()
            out[f"llm.{name}.out_mask"] = o.attention_mask.numpy()
        # last layer through output_key (no skip layer), all hidden states returned
        enc.hidden_state_skip_layer = None
        toks = enc.text2tokens("a red car", data_type="image")
        o = enc.encode(toks, data_type="image", output_hidden_states=True)
        out["llm.last.input_ids"] = toks["input_ids"].numpy()
        out["llm.last.attention_mask"] = toks["attention_mask"].numpy()
        out["llm.last.hidden_state"] = o.hidden_state.float().numpy()
        out["llm.last.n_hidden_states"] = np.array(len(o.hidden_states_list))
        # skip layer without the final norm
        enc.apply_final_norm = False
        o = enc.encode(toks, data_type="image", hidden_state_skip_layer=1)
        out["llm.skip1_nonorm.hidden_state"] = o.hidden_state.float().numpy()
        # ---------------- CLIP-L
        clip_dir = os.path.join(tmp, "clip")
        os.makedirs(clip_dir)
        nvocab = write_clip_tokenizer_files(clip_dir)
        assert nvocab <= CLIP_CFG["vocab_size"]
        torch.manual_seed(1)
        clip = CLIPTextModel(CLIPTextConfig(**CLIP_CFG)).eval()
        clip.save_pretrained(clip_dir)
        for k, v in clip.state_dict().items():
            out["clip.w." + k] = v.detach().float().numpy()
        try:
            enc2 = REF.TextEncoder("clipL", max_length=10, text_encoder_precision="fp32", text_encoder_path=clip_dir, tokenizer_type="clipL")
            note = "reference load_text_encoder ran as is"
        except AttributeError as e:
            # transformers 5.x holds the CLIP text tower directly on CLIPTextModel (no .text_model): the reference's loader line :35 is an
            # ordinary AttributeError under this library version; everything after the load (text2tokens / encode) is still reference code
            note = f"reference load_text_encoder raised {type(e).__name__} ({e}) under transformers {__import__('transformers').__version__}: " \
                   "model attached by hand, text2tokens/encode are the reference's"
            orig = REF.load_text_encoder

            def patched(text_encoder_type, text_encoder_precision=None, text_encoder_path=None, logger=None, device=None):
                m = CLIPTextModel.from_pretrained(text_encoder_path)
                m.final_layer_norm = getattr(m, "text_model", m).final_layer_norm
                m.requires_grad_(False)
                return m, text_encoder_path
            REF.load_text_encoder = patched
            enc2 = REF.TextEncoder("clipL", max_length=10, text_encoder_precision="fp32", text_encoder_path=clip_dir, tokenizer_type="clipL")
            REF.load_text_encoder = orig
        print("CLIP:", note)
        toks = enc2.text2tokens("a red car")
        o = enc2.encode(toks)
        out["clip.input_ids"] = toks["input_ids"].numpy()
        out["clip.attention_mask"] = toks["attention_mask"].numpy()
        out["clip.hidden_state"] = o.hidden_state.float().numpy()
        out["clip.out_mask"] = o.attention_mask.numpy()
    np.savez_compressed(OUT, **out)
    print(f"wrote {OUT}: {len(out)} arrays, {os.path.getsize(OUT) / 1024:.0f} KiB")
    for k in sorted(out):
        if ".w." not in k:
            print(" ", k, out[k].shape)


if __name__ == "__main__":
    main()
