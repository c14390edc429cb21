"""CPU: the TextEncoder wrapper (reference hyvideo/text_encoder/__init__.py) over tiny random Hugging Face
models built from configs and an in-memory tokenizer (no checkpoints or tokenizer files exist here): template application and
padding (text2tokens :220-268), hidden-state selection, final norm and instruction-token cropping (encode :270-339), CLIP pooled
output; and the video writer's frame composition (utils/file_utils.py:47-70)."""
import numpy as np
import pytest
import torch


def _toy_tokenizer(max_len):
    from tokenizers import Tokenizer, models, pre_tokenizers
    from transformers import PreTrainedTokenizerFast
    words = ["[PAD]", "[UNK]", "[EOS]", "describe", "the", "video", ":", "a", "cat", "walks", "on", "grass", "slowly", "red", "car"]
    tok = Tokenizer(models.WordLevel({w: i for i, w in enumerate(words)}, unk_token="[UNK]"))
    tok.pre_tokenizer = pre_tokenizers.Whitespace()
    return PreTrainedTokenizerFast(tokenizer_object=tok, pad_token="[PAD]", unk_token="[UNK]", eos_token="[EOS]", model_max_length=max_len)


def _llm(hidden=32):
    from transformers import LlamaConfig, LlamaModel
    torch.manual_seed(0)
    return LlamaModel(LlamaConfig(vocab_size=32, hidden_size=hidden, intermediate_size=64, num_hidden_layers=4, num_attention_heads=4,
                                  num_key_value_heads=4, max_position_embeddings=64, pad_token_id=0))


def test_llm_text_encoder_template_skip_layer_and_crop():
    from hunyuanvideo_efficiency_amd.text_encoder import TextEncoder
    tpl = {"template": "describe the video : {}", "crop_start": 4}
    text_len = 8
    enc = TextEncoder("llm", max_length=text_len + tpl["crop_start"], text_encoder_precision="fp32", tokenizer_type="llm",
                      prompt_template={"template": "describe : {}", "crop_start": 2}, prompt_template_video=tpl,
                      hidden_state_skip_layer=2, apply_final_norm=True, model=_llm(), tokenizer=_toy_tokenizer(64))
    assert enc.output_key == "last_hidden_state" and enc.model.final_layer_norm is enc.model.norm
    toks = enc.text2tokens("a cat walks on grass", data_type="video")
    assert toks["input_ids"].shape == (1, 12) and toks["attention_mask"][0].tolist() == [1] * 9 + [0] * 3
    assert toks["input_ids"][0, :4].tolist() == [3, 4, 5, 6]                 # template tokens in front
    out = enc.encode(toks, data_type="video")
    ref = enc.model(input_ids=toks["input_ids"], attention_mask=toks["attention_mask"], output_hidden_states=True)
    exp = enc.model.norm(ref.hidden_states[-3])[:, 4:]
    assert out.hidden_state.shape == (1, text_len, 32) and torch.allclose(out.hidden_state, exp, atol=1e-6)
    assert out.attention_mask.tolist() == [[1] * 5 + [0] * 3]
    # image template: other crop; no final norm; last layer via output_key when no skip layer
    enc.apply_final_norm = False
    o2 = enc.encode(enc.text2tokens("red car", data_type="image"), data_type="image", hidden_state_skip_layer=1)
    t2 = enc.text2tokens("red car", data_type="image")
    r2 = enc.model(input_ids=t2["input_ids"], attention_mask=t2["attention_mask"], output_hidden_states=True)
    assert torch.allclose(o2.hidden_state, r2.hidden_states[-2][:, 2:], atol=1e-6)
    enc.hidden_state_skip_layer = None
    o3 = enc.encode(t2, data_type="image", output_hidden_states=True)
    assert torch.allclose(o3.hidden_state, r2.last_hidden_state[:, 2:], atol=1e-6) and len(o3.hidden_states_list) == 5
    # list input, truncation to max_length, bad inputs
    tl = enc.text2tokens(["a cat", "a cat walks on grass slowly a cat walks on grass slowly"], data_type="video")
    assert tl["input_ids"].shape == (2, 12) and int(tl["attention_mask"][1].sum()) == 12
    with pytest.raises(TypeError):
        enc.text2tokens(3)
    with pytest.raises(ValueError):
        enc.text2tokens("a cat", data_type="audio")
    with pytest.raises(ValueError):
        TextEncoder("bert", max_length=4, model=_llm(), tokenizer=_toy_tokenizer(8))
    with pytest.raises(AssertionError):
        TextEncoder("llm", max_length=4, prompt_template={"template": "no placeholder"}, model=_llm(), tokenizer=_toy_tokenizer(8))


def test_clip_text_encoder_pooled_output():
    from transformers import CLIPTextConfig, CLIPTextModel
    from hunyuanvideo_efficiency_amd.text_encoder import TextEncoder
    torch.manual_seed(1)
    clip = CLIPTextModel(CLIPTextConfig(vocab_size=32, hidden_size=32, intermediate_size=64, num_hidden_layers=2, num_attention_heads=4,
                                        max_position_embeddings=16, projection_dim=32, pad_token_id=0, bos_token_id=1, eos_token_id=2))
    enc = TextEncoder("clipL", max_length=10, text_encoder_precision="fp32", model=clip, tokenizer=_toy_tokenizer(16))
    assert enc.output_key == "pooler_output" and enc.model.final_layer_norm is getattr(clip, "text_model", clip).final_layer_norm
    toks = enc.text2tokens("a red car [EOS]")
    out = enc.encode(toks)
    ref = clip(input_ids=toks["input_ids"], attention_mask=toks["attention_mask"])
    assert out.hidden_state.shape == (1, 32) and torch.allclose(out.hidden_state, ref.pooler_output, atol=1e-6)
    assert out.attention_mask.shape == (1, 10)


def test_pipeline_encode_prompt_glue():
    from hunyuanvideo_efficiency_amd.text_encoder import TextEncoder
    from hunyuanvideo_efficiency_amd.diffusion.pipelines import HunyuanVideoPipeline
    tpl = {"template": "describe the video : {}", "crop_start": 4}
    enc = TextEncoder("llm", max_length=12, text_encoder_precision="fp32", prompt_template=tpl, prompt_template_video=tpl,
                      hidden_state_skip_layer=2, model=_llm(), tokenizer=_toy_tokenizer(64))
    pipe = HunyuanVideoPipeline(None, None, None, text_encoder=enc)
    emb, neg, mask, negmask = pipe.encode_prompt("a cat walks", "cpu", 1, False, data_type="video")
    assert emb.shape == (1, 8, 32) and mask.tolist() == [[1, 1, 1, 0, 0, 0, 0, 0]] and neg is None and negmask is None
    # classifier-free guidance: the negative prompt ("" by default, pipeline_hunyuan_video.py:373-374) is encoded the same way
    emb_c, emb_u, mask_c, mask_u = pipe.encode_prompt("a cat walks", "cpu", 1, True, data_type="video")
    assert torch.equal(emb_c, emb) and emb_u.shape == emb.shape and mask_u.tolist() == [[0] * 8]
    _, emb_n, _, mask_n = pipe.encode_prompt("a cat walks", "cpu", 1, True, "red car", data_type="video")
    assert mask_n.tolist() == [[1, 1, 0, 0, 0, 0, 0, 0]] and not torch.equal(emb_n, emb_u)
    with pytest.raises(TypeError):
        pipe.encode_prompt("a cat", "cpu", 1, True, ["red car"], data_type="video")


def test_save_videos_grid_frames_and_container(tmp_path):
    from hunyuanvideo_efficiency_amd.utils.file_utils import frames_uint8, save_videos_grid
    v = torch.rand(2, 3, 4, 6, 8) * 2 - 1
    fr = frames_uint8(v, rescale=True, n_rows=2)
    assert len(fr) == 4 and fr[0].shape == (6 + 4, 2 * 8 + 6, 3) and fr[0].dtype == np.uint8
    exp = ((v[0, :, 0].permute(1, 2, 0) + 1) / 2).clamp(0, 1).mul(255).numpy().astype(np.uint8)
    assert np.array_equal(fr[0][2:8, 2:10], exp)
    one = frames_uint8(v[:1].clamp(0, 1), rescale=False)
    assert one[0].shape == (6, 8, 3)
    out = save_videos_grid(v, str(tmp_path / "sub" / "clip.mp4"), rescale=True, n_rows=2, fps=8)
    import os
    assert os.path.exists(out) and os.path.getsize(out) > 0
    if out.endswith(".gif"):
        from PIL import Image
        im = Image.open(out)
        assert im.n_frames == 4 and im.size == (2 * 8 + 6, 10)


def test_text_encoder_vs_reference_executed_golden(golden, tmp_path):
    """SURVEY.md 8f row 4 pinned: tests/golden/text_encoder.npz was produced by tools/make_golden_text.py, which IMPORTS the reference's
    hyvideo/text_encoder and runs ITS TextEncoder.text2tokens / encode (:220-339) on tiny random LLaMA / CLIP models saved with
    save_pretrained.  Here the same models are rebuilt from the weights the fixture carries and this repo's wrapper must return the
    same token ids, masks and hidden states: video and image templates with instruction-token cropping, hidden_state_skip_layer = 2
    with the final norm, skip layer without it, the plain last layer via output_key, truncation, CLIP's pooled output."""
    from transformers import CLIPTextConfig, CLIPTextModel, CLIPTokenizer, LlamaConfig, LlamaModel
    from hunyuanvideo_efficiency_amd.text_encoder import TextEncoder
    from tests.toy_text_models import CLIP_CFG, LLM_CFG, TPL_IMAGE, TPL_VIDEO, toy_llm_tokenizer, write_clip_tokenizer_files
    g = golden("text_encoder")
    llm = LlamaModel(LlamaConfig(**LLM_CFG)).eval()
    llm.load_state_dict({k[len("llm.w."):]: v for k, v in g.items() if k.startswith("llm.w.")}, strict=True)
    enc = TextEncoder("llm", max_length=8 + TPL_VIDEO["crop_start"], text_encoder_precision="fp32", tokenizer_type="llm",
                      prompt_template=TPL_IMAGE, prompt_template_video=TPL_VIDEO, hidden_state_skip_layer=2, apply_final_norm=True,
                      model=llm, tokenizer=toy_llm_tokenizer())
    for name, text, dt in (("video", "a cat walks on grass", "video"), ("image", "red car", "image"),
                           ("video_long", "a cat walks on grass slowly a cat walks on grass slowly", "video")):
        toks = enc.text2tokens(text, data_type=dt)
        assert torch.equal(toks["input_ids"], g[f"llm.{name}.input_ids"]) and torch.equal(toks["attention_mask"], g[f"llm.{name}.attention_mask"])
        o = enc.encode(toks, data_type=dt)
        torch.testing.assert_close(o.hidden_state, g[f"llm.{name}.hidden_state"], rtol=1e-5, atol=1e-6)
        assert torch.equal(o.attention_mask, g[f"llm.{name}.out_mask"])
    enc.hidden_state_skip_layer = None
    toks = enc.text2tokens("a red car", data_type="image")
    assert torch.equal(toks["input_ids"], g["llm.last.input_ids"])
    o = enc.encode(toks, data_type="image", output_hidden_states=True)
    torch.testing.assert_close(o.hidden_state, g["llm.last.hidden_state"], rtol=1e-5, atol=1e-6)
    assert len(o.hidden_states_list) == int(g["llm.last.n_hidden_states"])
    enc.apply_final_norm = False
    o = enc.encode(toks, data_type="image", hidden_state_skip_layer=1)
    torch.testing.assert_close(o.hidden_state, g["llm.skip1_nonorm.hidden_state"], rtol=1e-5, atol=1e-6)
    # CLIP-L: pooled output (the reference's loader line for CLIP is an AttributeError under transformers 5.x - the fixture's note;
    # its text2tokens / encode produced these vectors)
    clip = CLIPTextModel(CLIPTextConfig(**CLIP_CFG)).eval()
    clip.load_state_dict({k[len("clip.w."):]: v for k, v in g.items() if k.startswith("clip.w.")}, strict=True)
    write_clip_tokenizer_files(str(tmp_path))
    enc2 = TextEncoder("clipL", max_length=10, text_encoder_precision="fp32", tokenizer_type="clipL", model=clip,
                       tokenizer=CLIPTokenizer.from_pretrained(str(tmp_path), max_length=77))
    toks = enc2.text2tokens("a red car")
    assert torch.equal(toks["input_ids"], g["clip.input_ids"]) and torch.equal(toks["attention_mask"], g["clip.attention_mask"])
    o = enc2.encode(toks)
    torch.testing.assert_close(o.hidden_state, g["clip.hidden_state"], rtol=1e-5, atol=1e-6)
    assert torch.equal(o.attention_mask, g["clip.out_mask"])
