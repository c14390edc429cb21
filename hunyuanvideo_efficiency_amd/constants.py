"""Constant tables of the reference that the callers either side of the hot path need (hyvideo/constants.py): the decoder-only
text encoder's instruction templates (strings the shipped DiT was trained against - they are data, and must match byte for byte;
`crop_start` = number of template tokens in front of the user prompt under the LLaVA-LLaMA-3 tokenizer) and the dtype names."""
import torch

PRECISION_TO_TYPE = {"fp32": torch.float32, "fp16": torch.float16, "bf16": torch.bfloat16}

# hyvideo/constants.py:32-47
PROMPT_TEMPLATE_ENCODE = (
    "<|start_header_id|>system<|end_header_id|>\n\nDescribe the image by detailing the color, shape, size, texture, "
    "quantity, text, spatial relationships of the objects and background:<|eot_id|>"
    "<|start_header_id|>user<|end_header_id|>\n\n{}<|eot_id|>"
)
PROMPT_TEMPLATE_ENCODE_VIDEO = (
    "<|start_header_id|>system<|end_header_id|>\n\nDescribe the video by detailing the following aspects: "
    "1. The main content and theme of the video."
    "2. The color, shape, size, texture, quantity, text, and spatial relationships of the objects."
    "3. Actions, events, behaviors temporal relationships, physical movement changes of the objects."
    "4. background environment, light, style and atmosphere."
    "5. camera angles, movements, and transitions used in the video:<|eot_id|>"
    "<|start_header_id|>user<|end_header_id|>\n\n{}<|eot_id|>"
)
# hyvideo/constants.py:51-60
PROMPT_TEMPLATE = {
    "dit-llm-encode": {"template": PROMPT_TEMPLATE_ENCODE, "crop_start": 36},
    "dit-llm-encode-video": {"template": PROMPT_TEMPLATE_ENCODE_VIDEO, "crop_start": 95},
}
