"""MI355X-native fusion layers — drop-in for the reference's ``models/fusion_layers.py``.

Same class names, constructor ``(config)``, forward signatures, return keys and
``state_dict()`` keys/shapes as the reference (citations: reference file ``models/fusion_layers.py``),
but every dense contraction, attention, LayerNorm, residual sum and pooling runs in the
hand-written gfx950 kernels of ``libmmfusion.so`` through ``mmfusion.ops``:

  * parameters stay fp32 ``nn.Parameter``s (reference checkpoints load), re-homed into a flat
    arena with a bf16 shadow for the MFMA kernels and an fp32 gradient arena (``mmfusion.arena``);
  * activations are bf16 ``(B*T, d)`` row-major; fp32 accumulation / softmax / LN statistics;
  * independent sub-blocks are issued as grouped launches (six cross-modal blocks -> one launch
    per stage instead of six).

There is no CPU path: calling these modules with CPU tensors raises.  Dropout (reference
``fusion_dropout``) is applied in training mode at the reference's sites (attention probabilities, FFN
hidden, fusion outputs) with the build's own counter-based RNG; parity runs use p = 0 or eval mode.
"""
from __future__ import annotations

import math
import warnings
from typing import Dict, List, Optional, Sequence, Tuple

import torch
import torch.nn as nn
import torch.nn.functional as F

from mmfusion import arena as _arena
from mmfusion import ops
from mmfusion import small_ops as sops
from mmfusion.ops import AttnSpec, LinearSpec, W

_depth = 0          # >0 while inside an outer fusion forward: the arena was already ensured
import os as _os
_BRANCH_STREAM = _os.environ.get("MMF_HIER_STREAMS", "1") != "0"   # HierarchicalFusion: small branches beside MulT
_MULT_NESTED = _os.environ.get("MMF_MULT_NESTED", "0") == "1"      # A/B: the two groups also when MulT runs inside HierarchicalFusion
_INTERLEAVE = _os.environ.get("MMF_HIER_INTERLEAVE", "1") != "0"    # HierarchicalFusion: side branches issued BETWEEN MulT's stages
# Side-branch thunks of an enclosing HierarchicalFusion, run one per stage of the MulT cross blocks (`_tick`).  A captured graph's nodes
# are submitted in the order they were created: with the four (B, d)-row branches (~100 launches) created in front of MulT their forward
# ran BEFORE MulT's first GEMM and — autograd executes in reverse creation order — their backward AFTER MulT's last, 250 + 300 us of
# small kernels alone on the chip although they sit on their own stream (profiles/r04_step_timeline_hier.txt); created first-MulT-then-
# branches the same serial section moves to the other end.  Created between MulT's stages they are submitted, forward and backward,
# while MulT's chip-filling launches run.  None entries skip a stage.
_between: list = []


def _tick() -> None:
    if _between:
        fn = _between.pop(0)
        if fn is not None:
            fn()
_RECAST = _os.environ.get("MMF_RECAST_EACH_STEP", "0") == "1"        # fp32 -> bf16 weight cast in every training forward
_MULT_STREAMS = int(_os.environ.get("MMF_MULT_STREAMS", "2"))       # MulT's cross blocks as this many concurrent groups (1, 2, 3)
# How the two groups are cut (round 3): "modality" = by QUERY modality — {t<-a, t<-v} + the text self-attention on one
# stream, {a<-t, a<-v, v<-t, v<-a} + the audio / video self-attentions on the other, joined only in front of the pooled
# projections — so the self-attention section (in-projection, cores, their backward: ~400 us of single-stream,
# partly-filled launches per step in rounds 1-2) also runs two streams wide; "size" = rounds 1-2: {t<-a, a<-v, v<-t} /
# {t<-v, a<-t, v<-a}, joined after the cross blocks.
_MULT_GROUPING = _os.environ.get("MMF_MULT_GROUPING", "modality")


class _FusionBase(nn.Module):
    """Arena handling shared by all fusion modules."""

    def _precision(self) -> str:
        """"bf16" (default) or "fp32" — the parity mode of mmfusion.ops_f32 (north_star "1e-3 fp32"): set
        ``module.precision``, ``config.fusion_precision`` (dynamic attribute) or ``MMF_PRECISION``."""
        return (getattr(self, "precision", None) or getattr(getattr(self, "config", None), "fusion_precision", None)
                or _os.environ.get("MMF_PRECISION") or "bf16")

    def _enter(self):
        global _depth
        if _depth == 0:
            # The bf16 shadow is a cache of the fp32 masters: it is re-cast only when a parameter's version counter has
            # moved (an in-place optimiser update, load_state_dict, ...) — the fused AdamW writes masters and shadow in
            # one pass and needs no cast at all.  MMF_RECAST_EACH_STEP=1 restores the per-forward cast of every
            # training forward (what autocast does; round 1's behaviour).  A captured hipGraph freezes this decision:
            # a replayed step whose masters are modified from outside must call arena.refresh(force=True) itself.
            _arena.ensure(self, refresh=self.training and _RECAST)
            self._saved_precision = ops.set_precision(self._precision())
            if self.training:
                ops.begin_training_forward()        # new dropout masks for this step
        _depth += 1

    def _exit(self):
        global _depth
        _depth -= 1
        if _depth == 0:
            ops.set_precision(getattr(self, "_saved_precision", "bf16"))
            _cat3_memo.clear()
            arena = getattr(next(self.parameters(), None), "_mmf_arena", None)
            if arena is not None:
                arena.join()            # side-stream cast / zeroing never outlives the root forward

    def __call__(self, *args, **kwargs):
        self._enter()
        try:
            return super().__call__(*args, **kwargs)
        finally:
            self._exit()


def _p(module: nn.Module, p: float) -> float:
    """Effective dropout probability: nn.Dropout / nn.MultiheadAttention(dropout=p) act in training mode
    only.  The HIP path draws its masks from a stateless hash of a device-resident step counter
    (mmfusion.ops: dropout section), so masks differ from torch's Philox stream — same distribution,
    checked statistically (tests/test_dropout_gpu.py); parity runs use p = 0 or eval()."""
    return float(p) if (module.training and p > 0.0) else 0.0


def _as_rows(x: torch.Tensor) -> torch.Tensor:
    """(B, T, d) or (B, d), fp32 or bf16 -> contiguous bf16 (B*T, d)."""
    if not x.is_cuda:
        raise RuntimeError("mmfusion fusion layers run on the GPU only (no CPU fallback)")
    return ops.to_bf16(x.contiguous()).reshape(-1, x.shape[-1])


def synth_flat(out) -> List[torch.Tensor]:
    """Every tensor in a (nested) result dict."""
    if isinstance(out, torch.Tensor):
        return [out]
    if isinstance(out, dict):
        return [t for v in out.values() for t in synth_flat(v)]
    return []


_cat3_memo: Dict[tuple, Tuple[torch.Tensor, torch.Tensor]] = {}


def _cat3(t: torch.Tensor, a: torch.Tensor, v: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
    """(fp32 (B, 3d), bf16 (B, 3d)) of cat([t, a, v], -1), computed once per root forward: Early, Contrastive,
    Adaptive and Graph fusion all start from the same concatenation (reference :36, :350, :436, :255-264)."""
    if not t.is_cuda:
        raise RuntimeError("mmfusion fusion layers run on the GPU only (no CPU fallback)")
    key = (t.data_ptr(), a.data_ptr(), v.data_ptr(), t._version, a._version, v._version, tuple(t.shape))
    hit = _cat3_memo.get(key)
    if hit is None:
        c32 = sops.cat3(t, a, v)
        hit = (c32, ops.to_bf16(c32))
        if _depth > 0:                    # the memo lives for one root forward (cleared in _FusionBase._exit)
            _cat3_memo[key] = hit
    return hit


def _lin(layer: nn.Linear, relu: bool = False) -> LinearSpec:
    return LinearSpec(W(layer.weight), W(layer.bias) if layer.bias is not None else None, relu)


class _MHAParams(nn.Module):
    """Parameter container with nn.MultiheadAttention's names, shapes and initialisation
    (packed in_proj_weight (3d, d) xavier-uniform, zero biases; torch activation.py)."""

    def __init__(self, embed_dim: int, num_heads: int):
        super().__init__()
        if embed_dim % num_heads:
            raise ValueError("embed_dim must be divisible by num_heads")
        self.embed_dim, self.num_heads, self.head_dim = embed_dim, num_heads, embed_dim // num_heads
        self.in_proj_weight = nn.Parameter(torch.empty(3 * embed_dim, embed_dim))
        self.in_proj_bias = nn.Parameter(torch.zeros(3 * embed_dim))
        self.out_proj = nn.Linear(embed_dim, embed_dim)
        nn.init.xavier_uniform_(self.in_proj_weight)
        nn.init.zeros_(self.out_proj.bias)

    def q_spec(self) -> LinearSpec:
        d = self.embed_dim
        return LinearSpec(W(self.in_proj_weight, 0, d), W(self.in_proj_bias, 0, d))

    def kv_spec(self) -> LinearSpec:
        d = self.embed_dim
        return LinearSpec(W(self.in_proj_weight, d, 3 * d), W(self.in_proj_bias, d, 3 * d))

    def qkv_spec(self) -> LinearSpec:
        return LinearSpec(W(self.in_proj_weight), W(self.in_proj_bias))


# ------------------------------------------------------------------------------------------------
# grouped building blocks
# ------------------------------------------------------------------------------------------------
def _cross_blocks(blocks: Sequence["CrossModalTransformer"], qs: Sequence[torch.Tensor],
                  kvs: Sequence[torch.Tensor], B: int, Tqs: Sequence[int], Tks: Sequence[int],
                  p: float = 0.0, ress: Optional[Sequence[torch.Tensor]] = None) -> List[torch.Tensor]:
    """n independent CrossModalTransformer blocks (reference :202-211), one launch per stage.
    qs[i]: bf16 (B*Tq_i, d); kvs[i]: bf16 (B*Tk_i, d); ress[i] (default qs[i]): the query rows as the residual of
    :205 — a separate handle when the caller fans its input out (ops.fanout) to sum the input gradients in one pass."""
    proj = _cross_in_proj(blocks, qs, kvs)                           # [Q0, KV0, Q1, KV1, ...]
    _tick()
    att = _cross_attention(blocks, proj, B, Tqs, Tks, p)
    _tick()
    return _cross_tail(blocks, att, qs if ress is None else ress, p)


def _cross_in_proj(blocks, qs, kvs) -> List[torch.Tensor]:
    items = []
    for blk, q, kv in zip(blocks, qs, kvs):
        items.append((q, blk.attention.q_spec(), None))
        items.append((kv, blk.attention.kv_spec(), None))
    return ops.linear_group(items)


def _cross_attention(blocks, proj, B, Tqs, Tks, p) -> List[torch.Tensor]:
    mp0 = blocks[0].attention
    d, H, dh = mp0.embed_dim, mp0.num_heads, mp0.head_dim
    specs = [AttnSpec(B, Tqs[i], Tks[i], q=(2 * i, 0), k=(2 * i + 1, 0), v=(2 * i + 1, d))
             for i in range(len(blocks))]
    return ops.attention_group(specs, H, dh, proj, dropout_p=p)             # dropout(p) on the probabilities


def _cross_tail(blocks, att, qs, p) -> List[torch.Tensor]:
    pre1 = ops.linear_group([(att[i], _lin(blk.attention.out_proj), qs[i]) for i, blk in enumerate(blocks)])
    _tick()
    x = ops.layernorm_group([(pre1[i], blk.norm1.weight, blk.norm1.bias) for i, blk in enumerate(blocks)],
                            blocks[0].norm1.eps)
    _tick()
    pre2 = ops.ffn_residual_group([(x[i], blk.ffn[0], blk.ffn[3]) for i, blk in enumerate(blocks)], dropout_p=p)
    _tick()
    return ops.layernorm_group([(pre2[i], blk.norm2.weight, blk.norm2.bias) for i, blk in enumerate(blocks)],
                               blocks[0].norm2.eps)


def _self_attention_core(mhas: Sequence[_MHAParams], xs: Sequence[torch.Tensor], B: int, Ts: Sequence[int],
                         p: float = 0.0) -> List[torch.Tensor]:
    """packed QKV projection + fused attention of n independent self-attention MHAs; the
    out-projection is left to the caller."""
    d, H, dh = mhas[0].embed_dim, mhas[0].num_heads, mhas[0].head_dim
    qkv = ops.linear_group([(x, m.qkv_spec(), None) for m, x in zip(mhas, xs)])
    specs = [AttnSpec(B, Ts[i], Ts[i], q=(i, 0), k=(i, d), v=(i, 2 * d)) for i in range(len(mhas))]
    return ops.attention_group(specs, H, dh, qkv, dropout_p=p)


def _self_attention(mhas: Sequence[_MHAParams], xs: Sequence[torch.Tensor], B: int, Ts: Sequence[int],
                    p: float = 0.0) -> List[torch.Tensor]:
    """n independent self-attention MHAs without residual/LN (reference :161-163)."""
    att = _self_attention_core(mhas, xs, B, Ts, p)
    return ops.linear_group([(att[i], _lin(m.out_proj), None) for i, m in enumerate(mhas)])


# ------------------------------------------------------------------------------------------------
# a4  EarlyFusion  (reference :9-43)
# ------------------------------------------------------------------------------------------------
class EarlyFusion(_FusionBase):
    def __init__(self, config):
        super().__init__()
        self.config = config
        d = config.fusion_hidden_size
        self.fusion_layers = nn.Sequential(
            nn.Linear(3 * d, 2 * d), nn.ReLU(), nn.Dropout(config.fusion_dropout),
            nn.Linear(2 * d, d), nn.ReLU(), nn.Dropout(config.fusion_dropout))

    def forward(self, text_features, audio_features, video_features) -> torch.Tensor:
        p = _p(self, self.config.fusion_dropout)
        _, x = _cat3(text_features, audio_features, video_features)                    # bf16 (B, 3d), :36
        h = ops.linear(x, *_wb(self.fusion_layers[0]), relu=True, dropout_p=p)
        return ops.linear(h, *_wb(self.fusion_layers[3]), relu=True, out_f32=True, dropout_p=p)


def _wb(layer: nn.Linear):
    return W(layer.weight), (W(layer.bias) if layer.bias is not None else None)


# ------------------------------------------------------------------------------------------------
# a5  LateFusion  (reference :46-90) — three d->7 heads through the narrow-linear kernel (f32 masters, exact);
# the softmax over the three learned fusion weights and the weighted logit sum are three scalars of torch glue
# ------------------------------------------------------------------------------------------------
class LateFusion(_FusionBase):
    def __init__(self, config):
        super().__init__()
        self.config = config
        d, n = config.fusion_hidden_size, config.num_emotions
        self.text_classifier = nn.Linear(d, n)
        self.audio_classifier = nn.Linear(d, n)
        self.video_classifier = nn.Linear(d, n)
        self.fusion_weights = nn.Parameter(torch.ones(3) / 3)

    def forward(self, text_features, audio_features, video_features) -> Dict[str, torch.Tensor]:
        if not text_features.is_cuda:
            raise RuntimeError("mmfusion fusion layers run on the GPU only (no CPU fallback)")
        tl = sops.narrow_linear(text_features, self.text_classifier)             # d -> num_emotions heads: HIP
        al = sops.narrow_linear(audio_features, self.audio_classifier)
        vl = sops.narrow_linear(video_features, self.video_classifier)
        w = F.softmax(self.fusion_weights, dim=0)                                  # three scalars
        return {"fused_logits": w[0] * tl + w[1] * al + w[2] * vl, "text_logits": tl,
                "audio_logits": al, "video_logits": vl, "fusion_weights": w}


# ------------------------------------------------------------------------------------------------
# a1  CrossModalTransformer  (reference :182-211)
# ------------------------------------------------------------------------------------------------
class CrossModalTransformer(_FusionBase):
    def __init__(self, config):
        super().__init__()
        d = config.fusion_hidden_size
        self.dropout_p = config.fusion_dropout
        self.attention = _MHAParams(d, config.fusion_num_heads)
        self.norm1 = nn.LayerNorm(d)
        self.norm2 = nn.LayerNorm(d)
        self.ffn = nn.Sequential(nn.Linear(d, 4 * d), nn.ReLU(), nn.Dropout(config.fusion_dropout),
                                 nn.Linear(4 * d, d))

    def forward(self, query: torch.Tensor, key_value: torch.Tensor) -> torch.Tensor:
        B, Tq, d = query.shape
        Tk = key_value.shape[1]
        y = _cross_blocks([self], [_as_rows(query)], [_as_rows(key_value)], B, [Tq], [Tk], _p(self, self.dropout_p))[0]
        return ops.to_f32(y).reshape(B, Tq, d)


# ------------------------------------------------------------------------------------------------
# a2  MultimodalTransformer (MulT)  (reference :93-179)
# ------------------------------------------------------------------------------------------------
class MultimodalTransformer(_FusionBase):
    def __init__(self, config):
        super().__init__()
        self.config = config
        d, H = config.fusion_hidden_size, config.fusion_num_heads
        self.text_to_audio = CrossModalTransformer(config)
        self.text_to_video = CrossModalTransformer(config)
        self.audio_to_text = CrossModalTransformer(config)
        self.audio_to_video = CrossModalTransformer(config)
        self.video_to_text = CrossModalTransformer(config)
        self.video_to_audio = CrossModalTransformer(config)
        self.text_self_attn = _MHAParams(d, H)
        self.audio_self_attn = _MHAParams(d, H)
        self.video_self_attn = _MHAParams(d, H)
        self.final_fusion = nn.Sequential(nn.Linear(3 * d, d), nn.ReLU(), nn.Dropout(config.fusion_dropout))

    def forward(self, text_features, audio_features, video_features) -> Dict[str, torch.Tensor]:
        p = _p(self, self.config.fusion_dropout)
        if text_features.dim() == 2:                                            # reference :140-143
            text_features, audio_features, video_features = (
                text_features.unsqueeze(1), audio_features.unsqueeze(1), video_features.unsqueeze(1))
        B, Tt, d = text_features.shape
        Ta, Tv = audio_features.shape[1], video_features.shape[1]
        t, a, v = _as_rows(text_features), _as_rows(audio_features), _as_rows(video_features)
        blocks = [self.text_to_audio, self.text_to_video, self.audio_to_text, self.audio_to_video,
                  self.video_to_text, self.video_to_audio]
        # every modality's rows are used seven times (2 queries + their residuals, 2 key/value sources, the three-way sum
        # :156-158): fan them out so that the seven input-gradient contributions are summed by ONE kernel in backward
        tf, af, vf = ops.fanout_group([t, a, v], 7)
        qs, kvs = [tf[0], tf[1], af[0], af[1], vf[0], vf[1]], [af[2], vf[2], tf[2], vf[3], tf[3], af[3]]
        ress = [tf[4], tf[5], af[4], af[5], vf[4], vf[5]]
        Tqs, Tks = [Tt, Tt, Ta, Ta, Tv, Tv], [Ta, Tv, Tt, Tv, Tt, Ta]
        t, a, v = tf[6], af[6], vf[6]
        mhas = [self.text_self_attn, self.audio_self_attn, self.video_self_attn]
        streams = _MULT_STREAMS > 1 and (_depth == 1 or _MULT_NESTED) and t.is_cuda and not ops.fp32_mode()
        # (as the root module only: nested in HierarchicalFusion the branch stream already fills the holes, and a third
        # stream measured slower: hier-seq 2.90 -> 3.24 ms)

        def run_blocks(g):
            return _cross_blocks([blocks[i] for i in g], [qs[i] for i in g], [kvs[i] for i in g], B,
                                 [Tqs[i] for i in g], [Tks[i] for i in g], p, [ress[i] for i in g])

        if streams and _MULT_STREAMS == 2 and _MULT_GROUPING in ("modality", "tv_a"):
            # Two independent chains up to the pooled projections, cut by QUERY modality; autograd replays each node's backward
            # on its forward stream, so the backward is two chains wide as well.  "modality": text on the current stream, audio +
            # video on the side stream.  "tv_a" (MMF_MULT_GROUPING=tv_a): text + video | audio — tried because the audio + video stream finishes its
            # backward ~110 us after the text stream (its 30-row attention problems are one-wave chains): level, 2.163 vs 2.165 ms.
            mods = {"modality": ([0], [1, 2]), "tv_a": ([0, 2], [1])}[_MULT_GROUPING]
            xs3, Ts3 = [t, a, v], [Tt, Ta, Tv]

            def chain(ms):          # query modalities ms: their two cross blocks each, the three-way sums, the self-attentions
                outs = run_blocks([2 * m + j for m in ms for j in (0, 1)])
                es = ops.add3_group([(xs3[m], outs[2 * i], outs[2 * i + 1]) for i, m in enumerate(ms)])   # :156-158
                return _self_attention_core([mhas[m] for m in ms], es, B, [Ts3[m] for m in ms], p)
            main = torch.cuda.current_stream()
            side = ops.branch_stream(1)                     # stream 0 belongs to HierarchicalFusion's small branches
            side.wait_stream(main)
            with torch.cuda.stream(side):
                att_side = chain(mods[1])
            att_main = chain(mods[0])
            att = [None, None, None]
            for m, x in zip(mods[0], att_main):
                att[m] = x
            for m, x in zip(mods[1], att_side):
                att[m] = x
            main.wait_stream(side)
            for x in att_side:
                x.record_stream(main)
        else:
            if streams:
                # The six blocks are independent: as balanced groups on concurrent streams, one group's HBM- / latency-bound
                # launches (attention, LayerNorm, residual adds, the partly filled last round of every GEMM launch) run
                # beside another group's GEMMs.  Forward here; autograd replays each node's backward on its forward stream.
                groups = {2: ([0, 3, 4], [1, 2, 5]),           # {t<-a, a<-v, v<-t} / {t<-v, a<-t, v<-a}: one query size each
                          3: ([0, 4], [1, 5], [2, 3])}[_MULT_STREAMS]
                main = torch.cuda.current_stream()
                outs = [None] * 6
                sides = []
                for gi, g in enumerate(groups[1:]):
                    side = ops.branch_stream(1 + gi)
                    side.wait_stream(main)
                    with torch.cuda.stream(side):
                        r = run_blocks(g)
                    for i, x in zip(g, r):
                        outs[i] = x
                    sides.append((side, g))
                g = groups[0]
                for i, x in zip(g, run_blocks(g)):
                    outs[i] = x
                for side, g in sides:
                    main.wait_stream(side)
                    for i in g:
                        outs[i].record_stream(main)
                t_a, t_v, a_t, a_v, v_t, v_a = outs
            else:
                t_a, t_v, a_t, a_v, v_t, v_a = _cross_blocks(blocks, qs, kvs, B, Tqs, Tks, p, ress)    # :146-153
            et, ea, ev = ops.add3_group([(t, t_a, t_v), (a, a_t, a_v), (v, v_t, v_a)])         # :156-158, one launch
            att = _self_attention_core(mhas, [et, ea, ev], B, [Tt, Ta, Tv], p)
        # :161-168.  The self-attention outputs are only ever used through their mean over T, and the
        # out-projection is affine, so mean_t(out_proj(o_t)) == out_proj(mean_t o_t): pool the attention
        # output first and run the three out-projections on (B, d) instead of (B*T, d) rows — the same
        # arithmetic up to fp reassociation, minus 2*(Tt+Ta+Tv)*d^2 FLOP/sample forward and twice that backward.
        pooled_att = ops.meanpool_cat([att[0].view(B, Tt, d), att[1].view(B, Ta, d), att[2].view(B, Tv, d)])
        pf = ops.linear_group([(x, _lin(m.out_proj), None) for x, m in zip(sops.split3(pooled_att), mhas)],
                              out_f32=True, cat=True)                           # :171 (B, 3d) f32, written in place
        fused = ops.linear(pf, *_wb(self.final_fusion[0]), relu=True, out_f32=True, dropout_p=p)      # :172
        return {"fused_features": fused, "text_features": pf[:, :d], "audio_features": pf[:, d:2 * d],
                "video_features": pf[:, 2 * d:]}


# ------------------------------------------------------------------------------------------------
# a6  GraphFusion  (reference :214-291) — dense 3-node GAT (PyG GATConv semantics, parity unpinned)
# ------------------------------------------------------------------------------------------------
class _DenseGAT(nn.Module):
    """Parameters of one PyG ``GATConv(in, out, heads=4, concat=False)`` (names of PyG >= 2.5:
    ``lin.weight``, ``att_src``, ``att_dst``, ``bias``; 2.3/2.4 checkpoints' ``lin_src``/``lin_dst``
    are accepted on load).  Glorot initialisation like PyG."""
    heads = 4

    def __init__(self, in_channels: int, out_channels: int):
        super().__init__()
        self.in_channels, self.out_channels = in_channels, out_channels
        self.lin = nn.Linear(in_channels, self.heads * out_channels, bias=False)
        self.att_src = nn.Parameter(torch.empty(1, self.heads, out_channels))
        self.att_dst = nn.Parameter(torch.empty(1, self.heads, out_channels))
        self.bias = nn.Parameter(torch.zeros(out_channels))
        nn.init.xavier_uniform_(self.lin.weight)
        nn.init.xavier_uniform_(self.att_src)
        nn.init.xavier_uniform_(self.att_dst)

    def _load_from_state_dict(self, state_dict, prefix, *args, **kwargs):
        for old in ("lin_src.weight", "lin_dst.weight"):
            if prefix + old in state_dict:
                w = state_dict.pop(prefix + old)
                state_dict.setdefault(prefix + "lin.weight", w)
        super()._load_from_state_dict(state_dict, prefix, *args, **kwargs)

    def forward(self, x: torch.Tensor, B: int, p: float = 0.0, pool: bool = False):
        """x: bf16 (B*3, in) -> relu(GATConv(x)) as bf16 (B*3, out) rows; with ``pool`` also the mean over the three
        nodes, bf16 (B, out).  The linear map is an MFMA GEMM, everything else one fused kernel (csrc/small.hip
        gat3_*: scores, leaky-relu, 3-way softmax, dropout on alpha with p, aggregation, head mean, bias, ReLU)."""
        h = ops.linear(x, W(self.lin.weight), None, out_f32=True)                      # (B*3, heads*out) f32
        if ops.fp32_mode():                      # parity mode: the (B, 3)-node arithmetic as f32 torch glue
            return sops.gat3_f32(h, self.att_src, self.att_dst, self.bias, B, self.heads, pool=pool)
        return sops.gat3(h, self.att_src, self.att_dst, self.bias, B, self.heads, relu=True, pool=pool, dropout_p=p)


class GraphFusion(_FusionBase):
    def __init__(self, config):
        super().__init__()
        self.config = config
        d, G, L = config.fusion_hidden_size, config.graph_hidden_size, config.graph_num_layers
        # The reference builds every layer d -> G but feeds layer >= 2 the G-wide output of the layer
        # before (models/fusion_layers.py:223-232 vs :280-282), so it only runs when G == d or L == 1
        # (SURVEY.md fact 4).  Here layer l >= 1 takes G inputs — identical shapes to the reference
        # whenever the reference can run, and a working stack at the default config (G=256, d=512).
        self.gcn_layers = nn.ModuleList([_DenseGAT(d if l == 0 else G, G) for l in range(L)])
        self.node_type_embedding = nn.Embedding(3, d)
        self.output_projection = nn.Linear(G, d)

    def forward(self, text_features, audio_features, video_features) -> torch.Tensor:
        B = text_features.shape[0]
        c32, _ = _cat3(text_features, audio_features, video_features)
        if ops.fp32_mode():
            x = (c32.view(B, 3, -1) + self.node_type_embedding.weight).reshape(B * 3, -1)
        else:
            x = sops.stack3_embed(c32, self.node_type_embedding.weight)                # :255-264, bf16 (B*3, d)
        pg = _p(self, self.config.graph_dropout)
        pooled = None
        for l, layer in enumerate(self.gcn_layers):                                    # :280-282
            if l + 1 < len(self.gcn_layers):
                x = layer(x, B, pg)
            else:
                x, pooled = layer(x, B, pg, pool=True)                                 # :285-286 mean over the nodes
        return ops.linear(pooled, *_wb(self.output_projection), out_f32=True)


# ------------------------------------------------------------------------------------------------
# a7  ContrastiveFusion  (reference :294-375)
# ------------------------------------------------------------------------------------------------
class ContrastiveFusion(_FusionBase):
    def __init__(self, config):
        super().__init__()
        self.config = config
        self.temperature = config.contrastive_temperature
        d = config.fusion_hidden_size
        mk = lambda: nn.Sequential(nn.Linear(d, d), nn.ReLU(), nn.Linear(d, d // 2))
        self.text_projector, self.audio_projector, self.video_projector = mk(), mk(), mk()
        self.fusion_layer = nn.Sequential(nn.Linear(3 * d, d), nn.ReLU(), nn.Dropout(config.fusion_dropout))

    def forward(self, text_features, audio_features, video_features,
                compute_contrastive_loss: bool = False) -> Dict[str, torch.Tensor]:
        p = _p(self, self.config.fusion_dropout)
        d = text_features.shape[-1]
        _, cat = _cat3(text_features, audio_features, video_features)                  # bf16 (B, 3d)
        xs = list(sops.split3(cat))
        projs = [self.text_projector, self.audio_projector, self.video_projector]
        h = ops.linear_group([(x, _lin(p[0], relu=True), None) for x, p in zip(xs, projs)])
        z = ops.linear_group([(hh, _lin(p[2]), None) for hh, p in zip(h, projs)], out_f32=True)
        losses = {}
        if text_features.shape[0] <= sops.NCE_MAX_B:
            # :338-347, 361-375: L2-normalise the three projections and the three symmetric InfoNCE losses in ONE
            # kernel (csrc/small.hip nce_*); the per-rank batch is small, the B x B similarities live in LDS
            (tp, ap, vp), ls = sops.normalize_infonce(z, self.temperature, compute_contrastive_loss)
            if compute_contrastive_loss:
                losses = {"text_audio": ls[0], "text_video": ls[1], "audio_video": ls[2]}
        else:                                   # B > 64 per rank: the B x B InfoNCE on torch ops (GPU)
            tp, ap, vp = (F.normalize(t, dim=-1) for t in z)
            if compute_contrastive_loss:
                losses = {"text_audio": self.contrastive_loss(tp, ap),
                          "text_video": self.contrastive_loss(tp, vp),
                          "audio_video": self.contrastive_loss(ap, vp)}
        fused = ops.linear(cat, *_wb(self.fusion_layer[0]), relu=True, out_f32=True, dropout_p=p)
        return {"fused_features": fused, "text_proj": tp, "audio_proj": ap, "video_proj": vp,
                "contrastive_losses": losses}

    def contrastive_loss(self, z1: torch.Tensor, z2: torch.Tensor) -> torch.Tensor:
        """Symmetric InfoNCE over the (per-rank) batch, reference :361-375; B x B, fp32."""
        sim = torch.mm(z1, z2.t()) / self.temperature
        labels = torch.arange(z1.size(0), device=z1.device)
        return (F.cross_entropy(sim, labels) + F.cross_entropy(sim.t(), labels)) / 2


# ------------------------------------------------------------------------------------------------
# a8  AdaptiveFusion  (reference :378-452)
# ------------------------------------------------------------------------------------------------
class AdaptiveFusion(_FusionBase):
    def __init__(self, config):
        super().__init__()
        self.config = config
        d = config.fusion_hidden_size
        self.attention = _MHAParams(d, config.fusion_num_heads)
        self.text_transform = nn.Linear(d, d)
        self.audio_transform = nn.Linear(d, d)
        self.video_transform = nn.Linear(d, d)
        self.weight_predictor = nn.Sequential(nn.Linear(3 * d, d), nn.ReLU(), nn.Linear(d, 3), nn.Softmax(dim=-1))
        self.fusion_layer = nn.Sequential(nn.Linear(d, d), nn.ReLU(), nn.Dropout(config.fusion_dropout))

    def forward(self, text_features, audio_features, video_features) -> Dict[str, torch.Tensor]:
        p = _p(self, self.config.fusion_dropout)
        B, d = text_features.shape
        mp = self.attention
        H, dh = mp.num_heads, mp.head_dim
        _, cat = _cat3(text_features, audio_features, video_features)                  # bf16 (B, 3d)
        xs = list(sops.split3(cat))
        stacked = ops.linear_group([(x, _lin(l), None) for x, l in
                                    zip(xs, (self.text_transform, self.audio_transform, self.video_transform))],
                                   cat=True).view(B * 3, d)                            # (B,3,d) :427-429, no copy
        qkv = ops.linear(stacked, mp.qkv_spec().w, mp.qkv_spec().b)
        att = ops.attention_group([AttnSpec(B, 3, 3, q=(0, 0), k=(0, d), v=(0, 2 * d))], H, dh, [qkv], dropout_p=p)[0]
        attended = ops.linear(att, *_wb(mp.out_proj), out_f32=True).view(B, 3, d)
        attn_w = torch.empty((B, 3, 3), dtype=torch.float32, device=qkv.device)        # head-averaged weights, returned
        if H <= 16 and not ops.fp32_mode():                                            # for inspection only (no gradient)
            from mmfusion import lib as _lib
            _lib.check(_lib.load().mmf_adaptive_attn_weights(qkv.data_ptr(), attn_w.data_ptr(), B, H, dh, _lib.stream_ptr()))
        else:
            with torch.no_grad():
                q4 = qkv.detach().float().view(B, 3, 3, H, dh)
                sc = torch.einsum("bihd,bjhd->bhij", q4[:, :, 0], q4[:, :, 1]) / math.sqrt(dh)
                attn_w = F.softmax(sc, dim=-1).mean(dim=1)
        hp = ops.linear(cat, *_wb(self.weight_predictor[0]), relu=True, out_f32=True)
        # :436-443: d -> 3 logits, softmax, weighted sum of the attended modalities — one kernel (small.hip ada_*)
        if ops.fp32_mode():                      # parity mode: d -> 3 logits, softmax, weighted sum as f32 torch glue
            aw = F.softmax(F.linear(hp, self.weight_predictor[2].weight, self.weight_predictor[2].bias), dim=-1)
            weighted = (attended * aw.unsqueeze(-1)).sum(dim=1)
        else:
            weighted, aw = sops.adaptive_combine(hp, attended, self.weight_predictor[2].weight, self.weight_predictor[2].bias)
        fused = ops.linear(weighted, *_wb(self.fusion_layer[0]), relu=True, out_f32=True, dropout_p=p)
        return {"fused_features": fused, "attention_weights": attn_w, "adaptive_weights": aw}


# ------------------------------------------------------------------------------------------------
# a9  HierarchicalFusion  (reference :455-520)
# ------------------------------------------------------------------------------------------------
class HierarchicalFusion(_FusionBase):
    """Five branches + meta MLP.  With (B, d) inputs this is the reference semantics (*hier-ref*).
    With (B, T, d) inputs — which the reference cannot take — the MulT branch
    consumes the sequences and the other four branches their mean over T: the build-defined
    *hier-seq* composition of SURVEY.md section 8(d)."""

    def __init__(self, config):
        super().__init__()
        self.config = config
        d = config.fusion_hidden_size
        self.early_fusion = EarlyFusion(config)
        self.mult_fusion = MultimodalTransformer(config)
        self.graph_fusion = GraphFusion(config)
        self.contrastive_fusion = ContrastiveFusion(config)
        self.adaptive_fusion = AdaptiveFusion(config)
        self.meta_fusion = nn.Sequential(nn.Linear(5 * d, 2 * d), nn.ReLU(), nn.Dropout(config.fusion_dropout),
                                         nn.Linear(2 * d, d))

    def _branches(self, text_features, audio_features, video_features, compute_contrastive_loss: bool):
        """-> (early, mult, graph, con, ada): the five branch results (:486-500)"""
        seq = (text_features, audio_features, video_features)
        rows_only = text_features.dim() == 2       # (B, d) inputs (the reference's own case): MulT is as small as the branches
        if text_features.dim() == 3:
            d = text_features.shape[-1]
            pooled = ops.to_f32(ops.meanpool_cat([ops.to_bf16(x.contiguous()) for x in seq]))
            text_features, audio_features, video_features = pooled[:, :d], pooled[:, d:2 * d], pooled[:, 2 * d:]
        if _BRANCH_STREAM and text_features.is_cuda and not ops.fp32_mode():
            # The four (B, d)-row branches are ~100 latency-bound launches of a few workgroups each and are independent
            # of the MulT branch: they run on a second stream beside MulT's chip-filling kernels (forward here; autograd
            # replays each node's backward on its forward stream) and join before the meta MLP.
            main, side = torch.cuda.current_stream(), ops.branch_stream()
            _cat3(text_features, audio_features, video_features)        # shared cast issued before the fork
            side.wait_stream(main)
            res = {}

            def on_side(name, fn):
                def run():
                    with torch.cuda.stream(side):
                        res[name] = fn()
                return run
            tav = (text_features, audio_features, video_features)
            thunks = [on_side("ada", lambda: self.adaptive_fusion(*tav)),                       # behind MulT's in-projections
                      on_side("con", lambda: self.contrastive_fusion(*tav, compute_contrastive_loss)),      # ... attention cores
                      on_side("early", lambda: self.early_fusion(*tav)),                        # ... out-projections
                      None,                                                                      # (LayerNorm 1)
                      on_side("graph", lambda: self.graph_fusion(*tav))]    # ... the FFN: the longest branch chain (3 GAT layers)
            # backward runs beside the FFN's two dgrads, the first big launches of MulT's backward
            if _INTERLEAVE and not rows_only:       # (with (B, d) inputs every launch is a few us: interleaving only adds cross-stream
                _between[:] = thunks                # hand-offs — MELD-shaped step 1.02 -> 1.06 ms)
            else:
                for fn in (thunks[2], thunks[4], thunks[1], thunks[0]):
                    fn()
            try:
                mult = self.mult_fusion(*seq)
                while _between:                                         # (a MulT path without those stages)
                    _tick()
            finally:
                del _between[:]
            early, graph, con, ada = res["early"], res["graph"], res["con"], res["ada"]
            main.wait_stream(side)
            for t in [early, graph] + [v for dct in (con, ada) for v in synth_flat(dct)]:
                t.record_stream(main)                                   # produced on `side`, consumed on `main` from here on
        else:
            early = self.early_fusion(text_features, audio_features, video_features)
            mult = self.mult_fusion(*seq)
            graph = self.graph_fusion(text_features, audio_features, video_features)
            con = self.contrastive_fusion(text_features, audio_features, video_features, compute_contrastive_loss)
            ada = self.adaptive_fusion(text_features, audio_features, video_features)
        return early, mult, graph, con, ada

    def _meta_hidden(self, allf: torch.Tensor, p: float) -> torch.Tensor:
        """the meta MLP's hidden layer (:507-508): Linear(5d, 2d) + ReLU + Dropout on the concatenated branch outputs"""
        return ops.linear(allf, *_wb(self.meta_fusion[0]), relu=True, dropout_p=p)

    def forward(self, text_features, audio_features, video_features,
                compute_contrastive_loss: bool = False) -> Dict[str, torch.Tensor]:
        p = _p(self, self.config.fusion_dropout)
        early, mult, graph, con, ada = self._branches(text_features, audio_features, video_features, compute_contrastive_loss)
        allf = torch.cat([early, mult["fused_features"], graph, con["fused_features"],
                          ada["fused_features"]], dim=-1)                             # :503-506, f32 (B, 5d): narrowed by the linear itself
        if ops.fp32_mode():
            allf = _as_rows(allf)
        h = self._meta_hidden(allf, p)
        final = ops.linear(h, *_wb(self.meta_fusion[3]), out_f32=True)
        return {"fused_features": final, "early_features": early, "mult_features": mult["fused_features"],
                "graph_features": graph, "contrastive_features": con["fused_features"],
                "adaptive_features": ada["fused_features"],
                "contrastive_losses": con.get("contrastive_losses", {}),
                "attention_weights": ada.get("attention_weights"),
                "adaptive_weights": ada.get("adaptive_weights")}
