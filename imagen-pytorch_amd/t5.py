"""Text-encoder hook of `Imagen.sample(texts=...)` (SURVEY.md §8(f) NEXT-3; reference: t5.py:60-118, call site ip.py:2326-2332).

The T5 encoder is UPSTREAM of the sampling path (every BASELINE config feeds precomputed `text_embeds`); this module only gives
the hook the reference has — `imagen.encode_text(texts, return_attn_mask=True) -> (embeds [B, L, D] fp32, mask [B, L] bool)` —
a default implementation: the Hugging Face T5 encoder run by PyTorch-ROCm on the sampling device, from LOCAL files only (this
stack never touches the network).  Without the weights on disk it fails loudly and tells the caller to pass `text_embeds=` or to
install an encoder with `imagen.encode_text = my_encoder`.
"""
from __future__ import annotations

from typing import Dict, List, Tuple

import torch

MAX_LENGTH = 256   # t5.py:16

_LOADED: Dict[str, Tuple[object, object]] = {}


def _model_and_tokenizer(name: str):
    if name not in _LOADED:
        try:
            from transformers import T5EncoderModel, T5Tokenizer
            tok = T5Tokenizer.from_pretrained(name, model_max_length=MAX_LENGTH, local_files_only=True)
            enc = T5EncoderModel.from_pretrained(name, local_files_only=True).eval()
        except Exception as e:   # missing weights / sentencepiece model / transformers itself
            raise RuntimeError(
                f"text encoder '{name}' is not available from local files ({type(e).__name__}: {e}). Pass precomputed `text_embeds=` "
                "to sample(), or set `imagen.encode_text = fn` with fn(texts, return_attn_mask=True) -> (embeds, mask)") from e
        _LOADED[name] = (enc, tok)
    return _LOADED[name]


@torch.no_grad()
def t5_encode_text(texts: List[str], name: str, return_attn_mask: bool = False, device=None):
    """t5.py:106-118: tokenise (longest padding, truncation at 256), encode, zero the padded positions."""
    enc, tok = _model_and_tokenizer(name)
    if device is None:
        device = torch.device("cuda") if torch.cuda.is_available() else torch.device("cpu")
    enc.to(device)
    batch = tok(list(texts), return_tensors="pt", padding="longest", max_length=MAX_LENGTH, truncation=True)
    ids, mask = batch.input_ids.to(device), batch.attention_mask.to(device)
    hidden = enc(input_ids=ids, attention_mask=mask).last_hidden_state.float()
    mask = mask.bool()
    hidden = hidden.masked_fill(~mask[..., None], 0.)   # t5.py:103
    return (hidden, mask) if return_attn_mask else hidden
