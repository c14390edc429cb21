"""Text-encoder side of the pipeline: the reference's `TextEncoder` wrapper (hyvideo/text_encoder/__init__.py)
over Hugging Face `transformers` models - a decoder-only LLM whose hidden states (layer -(skip+1), instruction tokens cropped)
are the DiT's `text_states`, and CLIP-L whose pooled output is `text_states_2`.  The models run on ROCm through torch (no custom
kernels: they execute once per video, outside the denoise loop).  Same constructor arguments, methods and output fields as the
reference; additionally `model=` / `tokenizer=` accept already-built objects (there are no checkpoints or tokenizer files in this
environment, and tests build tiny random models from configs)."""
from __future__ import annotations

from dataclasses import dataclass
from typing import Optional, Tuple

import torch
import torch.nn as nn

from ..constants import PRECISION_TO_TYPE


def use_default(value, default):
    return value if value is not None else default


def load_text_encoder(text_encoder_type, text_encoder_precision=None, text_encoder_path=None, logger=None, device=None, model=None):
    """text_encoder/__init__.py:17-55.  `final_layer_norm` is attached the way the reference does (CLIP: text_model's, LLM: .norm)."""
    if model is None:
        if text_encoder_path is None:
            raise ValueError("text_encoder_path is required (no default checkpoint locations in this build)")
        if logger is not None:
            logger.info(f"Loading text encoder model ({text_encoder_type}) from: {text_encoder_path}")
        if text_encoder_type == "clipL":
            from transformers import CLIPTextModel
            model = CLIPTextModel.from_pretrained(text_encoder_path)
        elif text_encoder_type == "llm":
            from transformers import AutoModel
            model = AutoModel.from_pretrained(text_encoder_path, low_cpu_mem_usage=True)
        else:
            raise ValueError(f"Unsupported text encoder type: {text_encoder_type}")
    if text_encoder_type == "clipL":
        # transformers < 5 nests the text tower under .text_model; 5.x holds embeddings / encoder / final_layer_norm directly
        inner = getattr(model, "text_model", model)
        if getattr(model, "final_layer_norm", None) is not inner.final_layer_norm:
            model.final_layer_norm = inner.final_layer_norm
    elif text_encoder_type == "llm":
        model.final_layer_norm = model.norm
    else:
        raise ValueError(f"Unsupported text encoder type: {text_encoder_type}")
    model.eval()
    if text_encoder_precision is not None:
        model = model.to(dtype=PRECISION_TO_TYPE[text_encoder_precision])
    model.requires_grad_(False)
    if device is not None:
        model = model.to(device)
    return model, text_encoder_path


def load_tokenizer(tokenizer_type, tokenizer_path=None, padding_side="right", logger=None, tokenizer=None):
    """text_encoder/__init__.py:58-74."""
    if tokenizer is not None:
        tokenizer.padding_side = padding_side
        return tokenizer, tokenizer_path
    if tokenizer_path is None:
        raise ValueError("tokenizer_path is required (no default checkpoint locations in this build)")
    if tokenizer_type == "clipL":
        from transformers import CLIPTokenizer
        tokenizer = CLIPTokenizer.from_pretrained(tokenizer_path, max_length=77)
    elif tokenizer_type == "llm":
        from transformers import AutoTokenizer
        tokenizer = AutoTokenizer.from_pretrained(tokenizer_path, padding_side=padding_side)
    else:
        raise ValueError(f"Unsupported tokenizer type: {tokenizer_type}")
    return tokenizer, tokenizer_path


@dataclass
class TextEncoderModelOutput:
    """text_encoder/__init__.py:77-97."""
    hidden_state: torch.Tensor = None
    attention_mask: Optional[torch.Tensor] = None
    hidden_states_list: Optional[Tuple[torch.Tensor, ...]] = None
    text_outputs: Optional[list] = None


class TextEncoder(nn.Module):
    def __init__(self, text_encoder_type: str, max_length: int, text_encoder_precision: Optional[str] = None,
                 text_encoder_path: Optional[str] = None, tokenizer_type: Optional[str] = None, tokenizer_path: Optional[str] = None,
                 output_key: Optional[str] = None, use_attention_mask: bool = True, input_max_length: Optional[int] = None,
                 prompt_template: Optional[dict] = None, prompt_template_video: Optional[dict] = None,
                 hidden_state_skip_layer: Optional[int] = None, apply_final_norm: bool = False, reproduce: bool = False,
                 logger=None, device=None, model=None, tokenizer=None):
        super().__init__()
        self.text_encoder_type = text_encoder_type
        self.max_length = max_length
        self.precision = text_encoder_precision
        self.model_path = text_encoder_path
        self.tokenizer_type = tokenizer_type if tokenizer_type is not None else text_encoder_type
        self.tokenizer_path = tokenizer_path if tokenizer_path is not None else text_encoder_path
        self.use_attention_mask = use_attention_mask
        if prompt_template_video is not None:
            assert use_attention_mask is True, "Attention mask is True required when training videos."
        self.input_max_length = input_max_length if input_max_length is not None else max_length
        self.prompt_template = prompt_template
        self.prompt_template_video = prompt_template_video
        self.hidden_state_skip_layer = hidden_state_skip_layer
        self.apply_final_norm = apply_final_norm
        self.reproduce = reproduce
        self.logger = logger
        for name, tpl in (("prompt_template", prompt_template), ("prompt_template_video", prompt_template_video)):
            if tpl is not None:
                assert isinstance(tpl, dict) and "template" in tpl, f"`{name}` must be a dictionary with a key 'template', got {tpl}"
                assert "{}" in str(tpl["template"]), f"`{name}['template']` must contain a placeholder `{{}}` for the input text"
        self.use_template = prompt_template is not None
        self.use_video_template = prompt_template_video is not None
        if "t5" in text_encoder_type:
            self.output_key = output_key or "last_hidden_state"
        elif "clip" in text_encoder_type:
            self.output_key = output_key or "pooler_output"
        elif "llm" in text_encoder_type or "glm" in text_encoder_type:
            self.output_key = output_key or "last_hidden_state"
        else:
            raise ValueError(f"Unsupported text encoder type: {text_encoder_type}")
        self.model, self.model_path = load_text_encoder(self.text_encoder_type, self.precision, self.model_path, logger, device, model)
        self.dtype = self.model.dtype
        self.device = self.model.device
        self.tokenizer, self.tokenizer_path = load_tokenizer(self.tokenizer_type, self.tokenizer_path, "right", logger, tokenizer)

    def __repr__(self):
        return f"{self.text_encoder_type} ({self.precision} - {self.model_path})"

    @staticmethod
    def apply_text_to_template(text, template, prevent_empty_text=True):
        if isinstance(template, str):
            return template.format(text)
        raise TypeError(f"Unsupported template type: {type(template)}")

    def _template(self, data_type):
        if data_type == "image":
            return self.prompt_template
        if data_type == "video":
            return self.prompt_template_video
        raise ValueError(f"Unsupported data type: {data_type}")

    def text2tokens(self, text, data_type="image"):
        """text_encoder/__init__.py:220-268: template applied, then padded/truncated to max_length."""
        if self.use_template:
            tpl = self._template(data_type)["template"]
            if isinstance(text, (list, tuple)):
                text = [self.apply_text_to_template(t, tpl) for t in text]
            elif isinstance(text, str):
                text = self.apply_text_to_template(text, tpl)
            else:
                raise TypeError(f"Unsupported text type: {type(text)}")
        return self.tokenizer(text, return_length=False, return_overflowing_tokens=False, return_attention_mask=True,
                              truncation=True, max_length=self.max_length, padding="max_length", return_tensors="pt")

    @torch.no_grad()
    def encode(self, batch_encoding, use_attention_mask=None, output_hidden_states=False, do_sample=None,
               hidden_state_skip_layer=None, return_texts=False, data_type="image", device=None):
        """text_encoder/__init__.py:270-339."""
        device = self.model.device if device is None else device
        use_attention_mask = use_default(use_attention_mask, self.use_attention_mask)
        hidden_state_skip_layer = use_default(hidden_state_skip_layer, self.hidden_state_skip_layer)
        attention_mask = batch_encoding["attention_mask"].to(device) if use_attention_mask else None
        outputs = self.model(input_ids=batch_encoding["input_ids"].to(device), attention_mask=attention_mask,
                             output_hidden_states=output_hidden_states or hidden_state_skip_layer is not None)
        if hidden_state_skip_layer is not None:
            last_hidden_state = outputs.hidden_states[-(hidden_state_skip_layer + 1)]
            # the real last hidden state already has the final norm applied; only intermediate layers may need it
            if hidden_state_skip_layer > 0 and self.apply_final_norm:
                last_hidden_state = self.model.final_layer_norm(last_hidden_state)
        else:
            last_hidden_state = outputs[self.output_key]
        if self.use_template:                       # drop the instruction tokens, keep the user prompt
            crop_start = self._template(data_type).get("crop_start", -1)
            if crop_start > 0:
                last_hidden_state = last_hidden_state[:, crop_start:]
                attention_mask = attention_mask[:, crop_start:] if use_attention_mask else None
        if output_hidden_states:
            return TextEncoderModelOutput(last_hidden_state, attention_mask, outputs.hidden_states)
        return TextEncoderModelOutput(last_hidden_state, attention_mask)

    def forward(self, text, use_attention_mask=None, output_hidden_states=False, do_sample=False, hidden_state_skip_layer=None,
                return_texts=False):
        return self.encode(self.text2tokens(text), use_attention_mask=use_attention_mask, output_hidden_states=output_hidden_states,
                           do_sample=do_sample, hidden_state_skip_layer=hidden_state_skip_layer, return_texts=return_texts)
