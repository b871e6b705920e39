"""Frozen CLIP text encoder for the five sensor paragraphs (reference core/clip.py:15-59).

The encoder runs once per process and is never on the hot path, so it stays a Hugging Face ``CLIPTextModel`` on
PyTorch-ROCm (north star: "Host code stays in Python on PyTorch-ROCm for ... the frozen CLIP text encoder").
Differences from the reference, all forced by running without a network:

* ``version`` must be a local directory holding the tokenizer and model files
  (``openai/clip-vit-large-patch14`` downloaded elsewhere); nothing is ever fetched.
* ``build_text_embeddings`` encodes the five paragraphs once and ``save_text_embeddings`` writes the cache file
  that ``opt['model']['text_embeddings']`` / ``WavBEST(text_embeddings=path)`` read, so inference hosts need
  neither the CLIP weights nor ``transformers``.

    python -m tmdiff_amd.clip --clip /data/clip-vit-large-patch14 --out text_embeddings.pt
"""
import argparse
import os

import torch
import torch.nn as nn

from .prompts import PROMPT_TEXT


class AbstractEncoder(nn.Module):
    def encode(self, *args, **kwargs):
        raise NotImplementedError


class FrozenCLIPEmbedder(AbstractEncoder):
    """Same constructor arguments, ``LAYERS`` and output selection as the reference (core/clip.py:15-56)."""
    LAYERS = ["last", "pooled", "hidden"]

    def __init__(self, version, device="cuda", max_length=77, freeze=True, layer="pooled", layer_idx=2):
        super().__init__()
        if layer not in self.LAYERS:
            raise ValueError(f"layer must be one of {self.LAYERS}")
        if not os.path.isdir(str(version)):
            raise FileNotFoundError(
                f"CLIP weights directory {version!r} not found: pass a local copy of openai/clip-vit-large-patch14 "
                "(this package never downloads), or use a text-embedding cache file instead")
        from transformers import CLIPTextModel, CLIPTokenizer   # imported late: inference hosts do not need it
        self.tokenizer = CLIPTokenizer.from_pretrained(version, local_files_only=True)
        self.transformer = CLIPTextModel.from_pretrained(version, local_files_only=True)
        self.device = device
        self.max_length = max_length
        if freeze:
            self.freeze()
        self.transformer.to(device)
        self.layer = layer
        self.layer_idx = layer_idx
        if layer == "hidden":
            assert layer_idx is not None
            assert 0 <= abs(layer_idx) <= 12

    def freeze(self):
        self.transformer = self.transformer.eval()
        for p in self.parameters():
            p.requires_grad = False

    def forward(self, text):
        enc = self.tokenizer(text, truncation=True, max_length=self.max_length, return_length=True,
                             return_overflowing_tokens=False, padding="max_length", return_tensors="pt")
        tokens = enc["input_ids"].to(self.device)
        out = self.transformer(input_ids=tokens, output_hidden_states=self.layer == "hidden")
        if self.layer == "last":
            return out.last_hidden_state[:, -1]
        if self.layer == "pooled":
            return out.pooler_output
        return out.hidden_states[self.layer_idx]

    def encode(self, text):
        return self(text)


@torch.no_grad()
def build_text_embeddings(embedder):
    """prompt name -> pooled [1, D] fp32 CPU tensor, what WavBEST.encode_prompt computes (Hyper_unet_general.py:566-572)."""
    return {name: embedder.encode(text).detach().float().cpu().reshape(1, -1) for name, text in PROMPT_TEXT.items()}


def save_text_embeddings(path, table):
    torch.save({k: v.detach().float().cpu() for k, v in table.items()}, path)


def load_text_embeddings(path):
    table = torch.load(path, map_location="cpu")
    missing = [k for k in PROMPT_TEXT if k not in table]
    if missing:
        raise KeyError(f"{path}: no embedding for prompt(s) {missing}")
    return {k: table[k].float().reshape(1, -1) for k in PROMPT_TEXT}


def main(argv=None):
    ap = argparse.ArgumentParser(description="Encode the five sensor paragraphs once and write the cache file.")
    ap.add_argument("--clip", required=True, help="local directory with the CLIP tokenizer + text model")
    ap.add_argument("--out", required=True, help="output .pt file (dict prompt -> [1,768])")
    ap.add_argument("--device", default="cuda" if torch.cuda.is_available() else "cpu")
    a = ap.parse_args(argv)
    table = build_text_embeddings(FrozenCLIPEmbedder(a.clip, device=a.device))
    save_text_embeddings(a.out, table)
    print({k: tuple(v.shape) for k, v in table.items()})


if __name__ == "__main__":
    main()
