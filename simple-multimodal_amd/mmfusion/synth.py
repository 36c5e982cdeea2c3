"""Seeded synthetic features and parameters for the fusion path.

The reference's ``create_sample_data.py`` writes wav/mp4/CSV for its (out-of-scope) backbones and
emits no feature tensors; this is the build's counterpart for the fusion path: the synthetic
``(B, T, d)`` / ``(B, d)`` feature tensors of SURVEY.md section 8(d) and a deterministic parameter
generator, so that fixtures need only store seeds, not weights.

Everything is generated on the CPU with an explicit ``torch.Generator`` and is therefore
identical in the build container and on the GPU box (same image, same torch).
"""
from __future__ import annotations

import math
from typing import Dict, Iterable, Sequence, Tuple

import torch

# BASELINE.json configs
C2_SHAPES = dict(B=16, T_text=512, T_audio=400, T_frames=30, d=768, heads=8)
INPUT_SEED = 1234
WEIGHT_SEED = 0


def make_features(B: int, Ts: Sequence[int], d: int, seed: int = INPUT_SEED,
                  dtype=torch.float32) -> Tuple[torch.Tensor, ...]:
    """N(0,1) features.  ``Ts`` entries of 0 mean a 2-D ``(B, d)`` tensor (as-wired path)."""
    g = torch.Generator().manual_seed(seed)
    out = []
    for T in Ts:
        shape = (B, d) if T == 0 else (B, T, d)
        out.append(torch.randn(shape, generator=g, dtype=torch.float32).to(dtype))
    return tuple(out)


def _scale_for(name: str, shape: Tuple[int, ...]) -> Tuple[float, float]:
    """(mean, std) for one parameter, by role.  Biases and LN/embedding parameters are
    deliberately non-trivial so that every epilogue term is exercised."""
    leaf = name.rsplit(".", 1)[-1]
    if leaf in ("att_src", "att_dst"):
        return 0.0, 1.0 / math.sqrt(shape[-1])
    if "norm" in name and leaf == "weight":
        return 1.0, 0.1
    if leaf == "fusion_weights":
        return 1.0 / 3.0, 0.2
    if len(shape) >= 2:                       # weight matrices / embeddings: fan-in scaled
        return 0.0, 1.0 / math.sqrt(shape[-1])
    return 0.0, 0.05                          # biases


def make_params(shapes: Iterable[Tuple[str, Tuple[int, ...]]], seed: int = WEIGHT_SEED
                ) -> Dict[str, torch.Tensor]:
    """Deterministic fp32 parameters for a list of (state_dict key, shape), drawn in sorted-key
    order from one generator."""
    g = torch.Generator().manual_seed(seed)
    out: Dict[str, torch.Tensor] = {}
    for name, shape in sorted((k, tuple(s)) for k, s in shapes):
        mean, std = _scale_for(name, shape)
        out[name] = torch.randn(shape, generator=g, dtype=torch.float32) * std + mean
    return out


def probe_vector(name: str, numel: int, seed: int = 4321) -> torch.Tensor:
    """Fixed random probe for gradient checksums: fixtures store <grad, probe> and ||grad||
    for large parameters instead of the full gradient."""
    h = 0
    for ch in name:
        h = (h * 131 + ord(ch)) % 2147483647
    g = torch.Generator().manual_seed(seed + h)
    return torch.randn(numel, generator=g, dtype=torch.float32)


# ---------------------------------------------------------------------------------------
# a scalar loss that touches every differentiable output with a non-uniform cotangent
# ---------------------------------------------------------------------------------------
NON_DIFF_KEYS = ("attention_weights",)      # returned for inspection only (never in a reference loss)


def flatten_outputs(out, prefix: str = "") -> Dict[str, torch.Tensor]:
    """Flatten a module result (Tensor or nested dict of Tensors) into {dotted key: tensor}."""
    if isinstance(out, torch.Tensor):
        return {prefix or "output": out}
    flat: Dict[str, torch.Tensor] = {}
    for k in sorted(out.keys()):
        v = out[k]
        key = f"{prefix}.{k}" if prefix else k
        if isinstance(v, torch.Tensor):
            flat[key] = v
        elif isinstance(v, dict):
            flat.update(flatten_outputs(v, key))
    return flat


def probe_loss(out) -> torch.Tensor:
    """sum_k <out_k, probe_k>: one scalar whose gradient w.r.t. every output element is a fixed
    pseudo-random number, so transposed or permuted gradients cannot cancel."""
    total = None
    for k, v in flatten_outputs(out).items():
        if k.rsplit(".", 1)[-1] in NON_DIFF_KEYS or not v.is_floating_point():
            continue
        p = probe_vector("out:" + k, v.numel()).reshape(v.shape).to(device=v.device)
        term = (v.float() * p).sum()
        total = term if total is None else total + term
    return total
