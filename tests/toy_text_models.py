"""Toy text models shared by tools/make_golden_text.py (which runs the REFERENCE's TextEncoder on them) and tests/test_text_encoder_cpu.py
(which runs this repo's wrapper on the same models rebuilt from the fixture's weights): configs, templates and tokenizers.  Test
infrastructure only."""
import json
import os

WORDS = ["[PAD]", "[UNK]", "[EOS]", "describe", "the", "video", ":", "a", "cat", "walks", "on", "grass", "slowly", "red", "car", "image"]
TPL_VIDEO = {"template": "describe the video : {}", "crop_start": 4}
TPL_IMAGE = {"template": "describe the image : {}", "crop_start": 4}
LLM_CFG = dict(vocab_size=32, hidden_size=32, intermediate_size=64, num_hidden_layers=4, num_attention_heads=4, num_key_value_heads=4,
               max_position_embeddings=64, pad_token_id=0)
CLIP_CFG = dict(vocab_size=64, hidden_size=32, intermediate_size=64, num_hidden_layers=2, num_attention_heads=4, max_position_embeddings=16,
                projection_dim=32, pad_token_id=1, bos_token_id=0, eos_token_id=1)


def toy_llm_tokenizer(max_len=64):
    from tokenizers import Tokenizer, models, pre_tokenizers
    from transformers import PreTrainedTokenizerFast
    tok = Tokenizer(models.WordLevel({w: i for i, w in enumerate(WORDS)}, unk_token="[UNK]"))
    tok.pre_tokenizer = pre_tokenizers.Whitespace()
    return PreTrainedTokenizerFast(tokenizer_object=tok, pad_token="[PAD]", unk_token="[UNK]", eos_token="[EOS]", model_max_length=max_len)


def write_clip_tokenizer_files(d):
    """vocab.json + merges.txt of a CLIP-style BPE tokenizer over a tiny alphabet: every word is spelled out in characters, a few
    merges build whole-word tokens."""
    chars = list("acdegilnorstvw")
    vocab = {"<|startoftext|>": 0, "<|endoftext|>": 1}
    for c in chars:
        vocab[c] = len(vocab)
    for c in chars:
        vocab[c + "</w>"] = len(vocab)
    merges = [("c", "a"), ("ca", "t</w>"), ("r", "e"), ("re", "d</w>"), ("c", "ar</w>") if False else ("a", "r</w>"), ("c", "ar</w>")]
    for a, b in merges:
        vocab[a + b] = len(vocab)
    with open(os.path.join(d, "vocab.json"), "w") as f:
        json.dump(vocab, f)
    with open(os.path.join(d, "merges.txt"), "w") as f:
        f.write("#version: 0.2\n" + "\n".join(f"{a} {b}" for a, b in merges) + "\n")
    return len(vocab)


