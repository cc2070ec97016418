"""Text context producer: captions -> token ids -> rows of the LLM's embedding table, on the device.

Mirrors ``sampling_6d.py:121-137``: the reference loads the whole causal LM by name only to call
``llm.model.embed_tokens(tokens)``; the sampling path needs nothing but that ``[vocab, 4096]`` table.
Here the tokenizer and the table come from **local paths** (there is no network), the table stays
resident on the GPU (fp32 or a 16-bit dtype) and the lookup is ``t2p_op_embedding_gather``.

Tokenizer call, verbatim from the reference: ``tokenizer(list(captions), return_tensors="pt",
add_special_tokens=False, max_length=512, padding=True, truncation=True).input_ids`` -- pad to the
longest caption of the batch, truncate at 512, no BOS / EOS; the attention mask is not used, padded
positions embed the pad token.
"""
from __future__ import annotations

import json
import os

import torch

from . import _lib
from ._lib import T2PError, check, ptr, stream_ptr

TABLE_KEYS = ("model.embed_tokens.weight", "embed_tokens.weight", "weight")
_DT = {torch.float32: 0, torch.bfloat16: 1, torch.float16: 2}


def load_embedding_table(path: str) -> torch.Tensor:
    """The ``embed_tokens`` weight from a local file or Hugging Face checkpoint directory
    (``*.safetensors`` / ``*.bin`` / ``*.pt``, sharded or not).  Only that tensor is read."""
    if os.path.isdir(path):
        for idx in ("model.safetensors.index.json", "pytorch_model.bin.index.json"):
            f = os.path.join(path, idx)
            if os.path.isfile(f):
                wm = json.load(open(f))["weight_map"]
                for k in TABLE_KEYS:
                    if k in wm:
                        return load_embedding_table(os.path.join(path, wm[k]))
        for name in ("model.safetensors", "pytorch_model.bin", "embed_tokens.pt", "embed_tokens.safetensors"):
            f = os.path.join(path, name)
            if os.path.isfile(f):
                return load_embedding_table(f)
        raise FileNotFoundError(f"no checkpoint with an embed_tokens table under {path}")
    if path.endswith(".safetensors"):
        from safetensors import safe_open
        with safe_open(path, framework="pt", device="cpu") as f:
            for k in TABLE_KEYS:
                if k in f.keys():
                    return f.get_tensor(k)
        raise KeyError(f"{path}: none of {TABLE_KEYS}")
    obj = torch.load(path, map_location="cpu")
    if torch.is_tensor(obj):
        return obj
    for k in TABLE_KEYS:
        if k in obj:
            return obj[k]
    raise KeyError(f"{path}: none of {TABLE_KEYS}")


def load_tokenizer(path: str):
    """``LlamaTokenizer.from_pretrained(llm_name, use_fast=False)`` (sampling_6d.py:122) from a local directory."""
    from transformers import LlamaTokenizer
    tok = LlamaTokenizer.from_pretrained(path, use_fast=False, local_files_only=True)
    if tok.pad_token is None:          # vicuna ships pad = unk; a bare LLaMA tokenizer has none and padding=True would raise
        tok.pad_token = tok.unk_token
    return tok


def tokenize_captions(tokenizer, captions, max_length=512) -> torch.Tensor:
    """sampling_6d.py:134-136, verbatim keyword arguments."""
    enc = tokenizer(list(captions), return_tensors="pt", add_special_tokens=False, max_length=max_length, padding=True,
                    truncation=True)
    return enc.input_ids


class TextContextProducer:
    def __init__(self, tokenizer, table: torch.Tensor, device="cuda:0", table_dtype=None, max_length=512):
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise T2PError("TextContextProducer needs a GPU device (there is no CPU fallback)")
        self.tokenizer = tokenizer
        self.max_length = max_length
        if table.dim() != 2:
            raise ValueError("embedding table must be [vocab, dim]")
        dt = table_dtype or (table.dtype if table.dtype in _DT else torch.float32)
        self.table = table.to(self.device, dt).contiguous()
        self.lib = _lib.load()
        self._bad = torch.zeros(1, device=self.device, dtype=torch.int32)

    @classmethod
    def from_local(cls, tokenizer_path, table_path, **kw):
        return cls(load_tokenizer(tokenizer_path), load_embedding_table(table_path), **kw)

    def tokens(self, captions) -> torch.Tensor:
        return tokenize_captions(self.tokenizer, captions, self.max_length)

    def embed(self, tokens: torch.Tensor) -> torch.Tensor:
        """sampling_6d.py:137: ``(B, T)`` ids -> ``(B, T, dim)`` float32 on the device."""
        ids = tokens.to(self.device, torch.int32).contiguous()
        B, T = ids.shape
        vocab, dim = self.table.shape
        out = torch.empty(B, T, dim, device=self.device, dtype=torch.float32)
        self._bad.zero_()
        with torch.cuda.device(self.device):
            check(self.lib.t2p_op_embedding_gather(ptr(self.table), _DT[self.table.dtype], ptr(ids), ptr(out), B * T, dim, vocab,
                                                   ptr(self._bad), stream_ptr()))
        if int(self._bad.item()):
            raise IndexError(f"token id outside the embedding table (vocab {vocab})")
        return out

    def __call__(self, captions) -> torch.Tensor:
        return self.embed(self.tokens(captions))
