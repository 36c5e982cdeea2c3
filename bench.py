#!/usr/bin/env python3
"""bench.py — fusion fwd+bwd samples/s (BASELINE.json metric) on N MI355X of one node.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload mult|hier|train|meld] [--no-graph]
                    [--no-cpu-baseline] [--frozen-inputs]
    N > 1: either the driver's `python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...`
    or plain `python bench.py --gpus N`, which starts exactly that launcher as a CHILD process before anything in this
    process has touched the GPU and relays its output and exit code.

Workload (BASELINE.json configs[1]): MulT cross-modal attention,
bf16 storage / f32 accumulate, synthetic features text (16,512,768), audio (16,400,768),
video (16,30,768) ~ N(0,1) from seed 1234+rank (generated in fp32, cast to bf16 once at set-up), default-initialised weights from
torch.manual_seed(0), fusion_dropout = 0, loss = fused_features.sum().  The inputs REQUIRE GRAD (headline): in the
model they are the outputs of the encoder tails, so the step computes the twelve in-projection dgrads and the sums of
the per-use input gradients as well.  `--frozen-inputs` times the round-1 variant (no input gradients); at N = 1 the
default run reports that number too, as `frozen_inputs`, beside the headline.

One "step" = gradient-arena zeroing, forward, backward — and, for N > 1, the RCCL all-reduce (mean) of the flat
gradient arena, the path's one exchange step.  The bf16 weight shadow the MFMA kernels read is a cache of the fp32
masters: fwd+bwd does not change them, so no cast runs in the step (--recast-weights puts round 1's whole-arena cast,
~62 us, back into every step; in the train workload the fused AdamW kernel writes the shadow itself).
Inputs are resident in HBM before the timed region.  Steps are replayed from one captured
hipGraph unless --no-graph.  Rank 0 prints ONE JSON line.

roofline: HIP-event durations of every grouped GEMM / attention launch are collected in a
separate eager pass after the timed region (same process, same shapes).  `roofline` names the kernel label
with the largest time SUMMED over its launches of a step (the rule of rounds 1-2; round 3 had switched to the longest
single launch in commit 71e2ace — that number is still reported, as `roofline_longest_launch`, beside
`roofline_gemm_all`: every grouped-GEMM launch of the step together); achieved = algorithmic FLOPs / time.
N > 1: after the headline region the same process group also times the hierarchical-fusion TRAINING step (the workload
of the north_star's scaling target, BASELINE configs[3]) and attaches it as `scaling_train`.
--workload meld: BASELINE configs[4], the path the reference really runs (pooled (B, 768) features -> Linear(768, 512)
-> ModalityDropout -> hierarchical fusion at d = 512, T = 1 -> classifier -> CE(ls 0.1) + 0.1 contrastive -> clip +
AdamW); HBM / launch bound: its roofline is bytes (parameters, gradients, optimiser state streamed once) over time.
step_tflops = the algorithmic FLOPs of the GEMM / attention problems ACTUALLY launched in one step (summed per launch
in that pass: 2MNK per GEMM problem, 4 / 8 Tq Tk d per attention problem fwd / bwd) / step time.
cpu_baseline: oracle/ref_cpu.py (fp32, all host cores) timed on rank 0 at N = 1 on the same
workload: 1 warm-up + best of 2 steps.

Collectives and rank-0-only work: every rank leaves the process group (barrier, destroy_process_group) right after the
timed region and the max-over-ranks reduction; rank 0's per-kernel pass, CPU baseline and printing run AFTER that, so
no rank-0-only region can contain a collective the other ranks are not in (the cause of both round-1 hangs,
DESIGN.md section 6) — with the group gone, mmfusion.dp's exchanges are no-ops instead of deadlocks.
"""
import argparse
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.abspath(__file__))
for p in (REPO, os.path.join(REPO, "simple-multimodal_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)
os.environ.setdefault("MMFUSION_CONFIG_MKDIRS", "0")
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import torch  # noqa: E402

BF16_MFMA_PEAK_TFLOPS = 2500.0      # dense, MI355X_MICROARCH.md chip table
HBM_PEAK_GBS = 8000.0


def mult_flops_per_sample(Tt, Ta, Tv, d):
    """Algorithmic forward FLOPs of MulT per sample: 6 cross blocks
    20 Tq d^2 + 4 Tk d^2 + 4 Tq Tk d, 3 self blocks 8 T d^2 + 4 T^2 d, final 6 d^2."""
    pairs = [(Tt, Ta), (Tt, Tv), (Ta, Tt), (Ta, Tv), (Tv, Tt), (Tv, Ta)]
    cross = sum(20 * q * d * d + 4 * k * d * d + 4 * q * k * d for q, k in pairs)
    selfa = sum(8 * t * d * d + 4 * t * t * d for t in (Tt, Ta, Tv))
    return cross + selfa + 6 * d * d


def build(workload, device, rank, dropout=0.0, input_grads=True):
    import config as cfgmod
    from mmfusion import synth
    from models import fusion_layers as fl
    from models.multimodal_model import EmotionClassifier
    S = synth.C2_SHAPES
    cfg = cfgmod.ModelConfig()
    cfg.fusion_hidden_size, cfg.fusion_num_heads, cfg.fusion_dropout = S["d"], S["heads"], dropout
    cfg.graph_hidden_size, cfg.graph_num_layers, cfg.graph_dropout = S["d"], 3, dropout
    torch.manual_seed(synth.WEIGHT_SEED)
    if workload == "train":
        class FusionWithHead(fl._FusionBase):          # one arena over fusion + classifier head
            def __init__(self):
                super().__init__()
                self.fusion_layer = fl.HierarchicalFusion(cfg)
                self.classifier = EmotionClassifier(cfg)

            def forward(self, t, a, v, compute_contrastive_loss=False):
                out = dict(self.fusion_layer(t, a, v, compute_contrastive_loss=compute_contrastive_loss))
                out["emotion_logits"] = self.classifier(out["fused_features"])
                return out
        model = FusionWithHead()
    else:
        model = (fl.MultimodalTransformer if workload == "mult" else fl.HierarchicalFusion)(cfg)
    model = model.to(device).train()
    # features ~ N(0,1) generated in fp32 from the seed, then cast to bf16 — once, here: the
    # step's inputs are the bf16 tensors resident in HBM
    xs = [t.to(device).to(torch.bfloat16) for t in synth.make_features(S["B"], (S["T_text"], S["T_audio"], S["T_frames"]),
                                                                     S["d"], seed=synth.INPUT_SEED + rank)]
    if input_grads:            # the encoder tails upstream need d(loss)/d(features): bf16 leaves that require grad
        for x in xs:
            x.requires_grad_(True)
    return cfg, model, xs


def build_meld(device, rank, dropout=0.1):
    """BASELINE configs[4]: per rank 3 x randn(16, 768) encoder features (seed 1234 + rank) ->
    Linear(768, 512) + dropout -> ModalityDropout(0.1) -> HierarchicalFusion(d = 512, H = 8, G = 512, L = 3) at T = 1 ->
    EmotionClassifier -> 7 classes.  Reference: models/multimodal_model.py:95-101,130-134 (projection tails and glue),
    training/advanced_trainer.py:139-182 (loss, clip, optimiser)."""
    import config as cfgmod
    from mmfusion import ops, synth
    from models import fusion_layers as fl
    from models.encoders import ModalityDropout
    from models.multimodal_model import EmotionClassifier
    cfg = cfgmod.ModelConfig()
    cfg.fusion_hidden_size, cfg.fusion_num_heads, cfg.fusion_dropout = 512, 8, dropout
    cfg.graph_hidden_size, cfg.graph_num_layers, cfg.graph_dropout = 512, 3, dropout
    torch.manual_seed(synth.WEIGHT_SEED)

    class MeldShaped(fl._FusionBase):
        def __init__(self):
            super().__init__()
            self.config = cfg
            self.text_projection = torch.nn.Linear(768, 512)
            self.audio_projection = torch.nn.Linear(768, 512)
            self.video_projection = torch.nn.Linear(768, 512)
            self.modality_dropout = ModalityDropout(0.1)
            self.fusion_layer = fl.HierarchicalFusion(cfg)
            self.classifier = EmotionClassifier(cfg)

        def forward(self, t, a, v, compute_contrastive_loss=False):
            p = fl._p(self, cfg.fusion_dropout)
            items = [(x, fl._lin(lin), None) for x, lin in            # f32 (16, 768) rows: narrowed, projected and dropped out in one launch
                     ((t, self.text_projection), (a, self.audio_projection), (v, self.video_projection))]
            feats = ops.linear_group(items, out_f32=True, dropout_p=p)
            feats = self.modality_dropout(*feats, training=self.training)
            out = dict(self.fusion_layer(*feats, compute_contrastive_loss=compute_contrastive_loss))
            out["emotion_logits"] = self.classifier(out["fused_features"])
            return out
    model = MeldShaped().to(device).train()
    g = torch.Generator().manual_seed(synth.INPUT_SEED + rank)
    xs = [torch.randn(16, 768, generator=g).to(device) for _ in range(3)]
    return cfg, model, xs


def make_train_step(model, xs, arena, world, allreduce, rank, shard_optimizer=False):
    """hierarchical-fusion TRAINING step (BASELINE configs[3]): zero grads, forward, CE(ls=0.1) + 0.1 *
    contrastive, backward | RCCL all-reduce | clip(1.0) + AdamW (fused, also refreshes the bf16 shadow)."""
    from mmfusion import dp
    from mmfusion.train import FusedAdamW, backward_from, fusion_loss, one_cycle_lr
    # --shard-optimizer (N > 1): ZeRO-1 step — reduce-scatter of the gradients, AdamW on this rank's 1/N of the arena,
    # all-gather of the bf16 shadow — instead of all-reduce + replicated AdamW (mmfusion.train.FusedAdamW.launch_sharded)
    opt = FusedAdamW(arena, lr=1e-4, weight_decay=1e-5, max_grad_norm=1.0, shard=shard_optimizer)
    g = torch.Generator().manual_seed(99 + rank)
    labels = torch.randint(0, 7, (xs[0].shape[0],), generator=g).to(xs[0].device)

    opt.set_schedule(1e-4, 100000)       # OneCycleLR evaluated on the device by opt.advance(), inside the captured step

    def fwd_bwd():
        arena.zero_grad(overlap=True, lazy=True)
        for x in xs:
            x.grad = None                # input gradients are produced anew each step, not accumulated across steps
        out = model(*xs, compute_contrastive_loss=True)
        backward_from(fusion_loss(out, labels))
        arena.finalize_grads()

    def before_replay():                 # nothing crosses the host per step any more (pinned-buffer race)
        pass

    def opt_launch():
        opt.advance()
        opt.launch()

    def exchange():
        if world > 1 and not opt.sharded:
            dp.allreduce_grads(arena, compress=None if allreduce == "fp32" else "bf16")
    EXTRAS["opt_launch"], EXTRAS["opt"] = opt_launch, opt
    return fwd_bwd, before_replay, exchange, opt_launch, opt.sharded


EXTRAS = {}


def meld_report(line, arena, ms_per_step, extras):
    """BASELINE configs[4] is HBM / launch bound (every GEMM has 16 rows): its roofline is bytes over time.
    Algorithmic bytes per parameter and step: bf16 weight read in forward (2) and in the input-gradient pass (2), f32 gradient
    written (4); clip + AdamW reads gradient, master, both moments (16) and writes master, both moments and the bf16 shadow (14):
    38 B.  `roofline` = the dominant kernel, the fused AdamW pass (30 B / parameter), timed here with HIP events on its own."""
    opt_launch = extras.get("opt_launch")
    n = arena.numel
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for _ in range(3):
        opt_launch()
    e0.record()
    reps = 20
    for _ in range(reps):
        opt_launch()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / reps
    ach = 30.0 * n / (us * 1e-6) / 1e9
    line["metric"] = "MELD-shaped fusion training step samples/sec at B=16 d=512"
    line["roofline"] = {"bound": "hbm", "kernel": "sqnorm_kernel + adamw_step_kernel (clip + AdamW over the flat arenas)",
                        "achieved": round(ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 4),
                        "traffic": None, "avg_launch_us": round(us, 2), "parameters": n,
                        "algorithmic_bytes": "30 B / parameter: gradient, master, exp_avg, exp_avg_sq read; master, both moments, "
                                             "bf16 shadow written"}
    step_bytes = 38.0 * n
    line["roofline_step"] = {"bound": "hbm", "achieved": round(step_bytes / (ms_per_step * 1e-3) / 1e9, 1), "peak": HBM_PEAK_GBS,
                             "unit": "GB/s", "frac": round(step_bytes / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                             "algorithmic_bytes": "38 B / parameter and step (weights read twice as bf16, f32 gradient written, AdamW 30)"}
    line.pop("roofline_longest_launch", None)
    line.pop("roofline_gemm_all", None)


def make_step(workload, model, xs, arena):
    ones = {}

    def step():
        # gradients are zeroed lazily: vectors by memset, matrices by the first wgrad GEMM of the step overwriting
        # (ParamArena.zero_grad(lazy=True); finalize_grads() zeroes any matrix no wgrad wrote)
        arena.zero_grad(overlap=True, lazy=True)
        for x in xs:
            x.grad = None                # input gradients are produced anew each step, not accumulated across steps
        if workload == "mult":
            out = model(*xs)
            fused = out["fused_features"]
            # the loss VALUE feeds nothing on the device: it is reduced on the branch stream, beside the backward's first launches, and
            # joined at the end of the step (one graph node less on the single-stream middle of the step)
            if fused.is_cuda:
                from mmfusion import ops
                main, side = torch.cuda.current_stream(), ops.branch_stream()
                side.wait_stream(main)
                with torch.cuda.stream(side):
                    loss = fused.sum()
            else:
                loss = fused.sum()
            # d(sum)/d(fused) is a tensor of ones: hand autograd a resident one instead of letting loss.backward() build it
            # with a fill and an expand kernel per step (two graph nodes on the step's single-stream critical path)
            if "g" not in ones:
                ones["g"] = torch.ones_like(fused)
            torch.autograd.backward([fused], [ones["g"]])
            if fused.is_cuda:
                main.wait_stream(side)
                loss.record_stream(main)
        else:
            out = model(*xs, compute_contrastive_loss=True)
            fused, aux = out["fused_features"], list(out["contrastive_losses"].values())
            loss = fused.sum() + 0.1 * sum(aux)
            # as above (round 4): d(loss)/d(fused) = ones and d(loss)/d(each contrastive term) = 0.1 are resident tensors; the loss value
            # is still computed, its backward's expand / multiply launches (five graph nodes of the single-stream middle) are not
            if "g" not in ones:
                ones["g"] = torch.ones_like(fused)
                ones["a"] = [torch.full_like(a, 0.1) for a in aux]
            torch.autograd.backward([fused] + aux, [ones["g"]] + ones["a"])
        arena.finalize_grads()
        return loss
    return step


def host_cores():
    """CPU threads this process may really use: affinity mask, cgroup quota, and the GPU box's
    documented per-GPU share (16) — os.cpu_count() reports the whole 256-thread host."""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except (AttributeError, OSError):
        pass
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            quota, period = f.read().split()
            if quota != "max":
                n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return max(1, min(n, int(os.environ.get("MMF_CPU_BASELINE_THREADS", "16"))))


def cpu_baseline(workload):
    """oracle (fp32 CPU restatement) on the same workload; returns dict for the JSON line."""
    from mmfusion import synth
    from oracle import ref_cpu
    import config as cfgmod
    from models import fusion_layers as fl
    S = synth.C2_SHAPES
    cfg = cfgmod.ModelConfig()
    cfg.fusion_hidden_size, cfg.fusion_num_heads, cfg.fusion_dropout = S["d"], S["heads"], 0.0
    torch.manual_seed(synth.WEIGHT_SEED)
    m = fl.MultimodalTransformer(cfg)                          # parameter container only (CPU, never called)
    P = {k: v.detach().clone().requires_grad_(True) for k, v in m.state_dict().items()}
    xs = synth.make_features(S["B"], (S["T_text"], S["T_audio"], S["T_frames"]), S["d"])
    cores = host_cores()
    torch.set_num_threads(cores)
    best = float("inf")
    for i in range(3):
        for p in P.values():
            p.grad = None
        t0 = time.perf_counter()
        out = ref_cpu.multimodal_transformer(P, "", *xs, S["heads"])
        out["fused_features"].sum().backward()
        dt = time.perf_counter() - t0
        if i > 0:
            best = min(best, dt)
    model_name = "unknown"
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    model_name = line.split(":", 1)[1].strip()
                    break
    except OSError:
        pass
    return {"value": round(S["B"] / best, 3), "unit": "samples/s", "cores": cores, "kind": "port",
            "sample": f"oracle/ref_cpu.py MulT fwd+bwd fp32, B=16 T=512/400/30 d=768, 1 warm-up + best of 2 "
                      f"steps ({best:.2f} s/step) on {model_name}"}


def cpu_baseline_meld(model):
    """the MELD-shaped step on the oracle (fp32 CPU restatement, dropout 0): projections + hier-ref fusion + classifier +
    loss, forward and backward, and torch's AdamW step over the same parameters; best of 3 after one warm-up."""
    from oracle import ref_cpu
    cores = host_cores()
    torch.set_num_threads(cores)
    P = {k: v.detach().float().cpu().clone().requires_grad_(True) for k, v in model.state_dict().items()}
    g = torch.Generator().manual_seed(1234)
    xs = [torch.randn(16, 768, generator=g) for _ in range(3)]
    labels = torch.randint(0, 7, (16,), generator=torch.Generator().manual_seed(99))
    opt = torch.optim.AdamW(list(P.values()), lr=1e-4, weight_decay=1e-5)
    best = float("inf")
    for i in range(4):
        opt.zero_grad(set_to_none=True)
        t0 = time.perf_counter()
        t, a, v = (ref_cpu.linear(x, P[f"{m}_projection.weight"], P[f"{m}_projection.bias"]) for x, m in zip(xs, ("text", "audio", "video")))
        fo = ref_cpu.hierarchical_fusion(P, "fusion_layer.", t, a, v, num_heads=8, graph_num_layers=3, temperature=0.07,
                                         compute_contrastive_loss=True)
        logits = ref_cpu.emotion_classifier(P, "classifier.", fo["fused_features"])
        loss = torch.nn.functional.cross_entropy(logits, labels, label_smoothing=0.1) + 0.1 * sum(fo["contrastive_losses"].values())
        loss.backward()
        torch.nn.utils.clip_grad_norm_([p for p in P.values() if p.grad is not None], 1.0)
        opt.step()
        dt = time.perf_counter() - t0
        if i > 0:
            best = min(best, dt)
    return {"value": round(16 / best, 2), "unit": "samples/s", "cores": cores, "kind": "port",
            "sample": f"oracle/ref_cpu.py MELD-shaped training step fp32 (dropout 0), B=16, best of 3 steps ({best * 1e3:.1f} ms/step)"}


_SYMBOL = {"gemm2_grouped_kernel<NT,bf16>": ["gemm2_grouped_kernel<false, false, false>"],
           "gemm2_grouped_kernel<NT,f32>": ["gemm2_grouped_kernel<false, false, true>"],
           "gemm2_grouped_kernel<NN,bf16>": ["gemm2_grouped_kernel<false, true, false>"],
           "gemm2_grouped_kernel<TN,f32>": ["gemm2_grouped_kernel<true, true, true>"],
           "gemm6_grouped_kernel<TN,f32>": ["gemm6_grouped_kernel<true, true, 32, 4, true>"],
           "gemm6_grouped_kernel<NT,bf16>": ["gemm6_grouped_kernel<false, false, 32, 4, false>"],
           "gemm6_grouped_kernel<NN,bf16>": ["gemm6_grouped_kernel<false, true, 32, 4, false>"],
           "gemm7_grouped_kernel<NT,bf16>": ["gemm7_persistent_kernel<false, 0>", "gemm7_persistent_kernel<false, 1>",
                                             "gemm7_persistent_kernel<false, 3>", "gemm7_persistent_kernel<false, 9>", "gemm7_persistent_kernel<false, 67>"],
           "gemm7_grouped_kernel<NN,bf16>": ["gemm7_persistent_kernel<true, 0>", "gemm7_persistent_kernel<true, 4>",
                                             "gemm7_persistent_kernel<true, 8>"],
           "attn_fwd_kernel<96>": ["attn_fwd2n_kernel<96, false>", "attn_fwd2_kernel<96, false, 2>", "attn_fwd2_kernel<96, false>"],
           "attn_bwd_kernels<96>": ["attn_bwd_dq2_kernel<96, false>", "attn_bwd_dkv2_kernel<96, false>"]}


def pmc_lookup(label):
    """(HBM bytes per launch, MFMA utilisation, source file) of a bench kernel label from the committed PMC passes
    (profiles/*_pmc_traffic.json, written by tools/pmc_summary.py from rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE /
    SQ_VALU_MFMA_BUSY_CYCLES runs of this same workload, FETCH_SIZE x2 for gfx950).  bench.py cannot profile
    itself, hence the file — the JSON line names it as the `source` of these two fields; (None, None, None) if
    absent.  A label that covers two kernel generations (NT/NN: 256x256 and 256x128 tiles) reports the
    launch-weighted mean; the two backward attention kernels are summed."""
    import glob
    files = sorted(glob.glob(os.path.join(REPO, "profiles", "*_pmc_traffic.json")))
    if not files or label not in _SYMBOL:
        return None, None, None
    src = os.path.relpath(files[-1], REPO)
    try:
        with open(files[-1]) as f:
            K = json.load(f)["kernels"]
        hits = [K[s] for s in _SYMBOL[label] if s in K]
        if not hits:
            return None, None, None
        if label.startswith("attn_bwd"):
            return int(sum(h["hbm_bytes_per_launch"] for h in hits)), None, src
        n = sum(h["launches"] for h in hits)
        traffic = int(sum(h["hbm_bytes_per_launch"] * h["launches"] for h in hits) / n)
        utils = [(h["mfma_util"], h["launches"]) for h in hits if "mfma_util" in h]
        util = round(sum(u * c for u, c in utils) / sum(c for _, c in utils), 4) if utils else None
        return traffic, util, src
    except (OSError, ValueError, KeyError):
        return None, None, None


def kernel_profile(step, nsteps):
    """Eager pass with HIP events around every grouped GEMM / attention launch."""
    from mmfusion import lib
    lib.PROFILE = []
    for _ in range(nsteps):
        step()
    torch.cuda.synchronize()
    from mmfusion import synth
    S = synth.C2_SHAPES
    agg = {}
    for label, flops, e0, e1, detail in lib.PROFILE:
        a = agg.setdefault(label, [0.0, 0.0, 0, 0.0])
        a[0] += e0.elapsed_time(e1)
        a[1] += flops
        a[2] += 1
        if label.startswith("attn_fwd") and detail:      # min HBM bytes of a core = bf16 Q, K, V in, O out
            a[3] += sum(2.0 * (2 * tq + 2 * tk) * S["d"] * S["B"] for tq, tk in detail)
    lib.PROFILE = None
    return {k: {"ms_total": v[0], "flops_total": v[1], "launches": v[2], "bytes_total": v[3]} for k, v in agg.items()}


def free_port():
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def self_launch(n, argv):
    """`python bench.py --gpus N` outside a launcher: start N ranks with torch.distributed.run as a CHILD process and
    relay its exit code (rank 0's JSON line goes straight to the inherited stdout).  Called before this process has
    made any GPU call (torch.cuda.device_count() does not initialise the GPU on this image); the launcher is never
    exec'ed into a process that has touched the GPU."""
    import subprocess
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr",
           "127.0.0.1", "--master-port", str(free_port()), os.path.abspath(__file__)] + list(argv)
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.run(cmd, env=env).returncode


def timed_region(run_step, steps, warmup, world, dist=None, sync=lambda: None, reduce_device="cpu"):
    """The contract's timed region: W untimed steps, then exactly K steps bracketed by barrier + device sync on both
    sides; returns the MAX over ranks of the elapsed seconds.  Every rank runs this, collectives included."""
    for _ in range(warmup):
        run_step()
    if world > 1:
        dist.barrier()
    sync()
    t0 = time.perf_counter()
    for _ in range(steps):
        run_step()
    sync()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], device=reduce_device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    return elapsed


def leave_group(world, dist=None):
    """All ranks: barrier, then destroy the process group.  Whatever runs after this (rank 0's per-kernel pass, CPU
    baseline, printing) cannot contain a collective: torch.distributed is no longer initialised, so mmfusion.dp's
    exchanges see a world of one and return instead of waiting for ranks that are not there."""
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--workload", choices=["mult", "hier", "train", "meld"], default="mult",
                    help="mult: MulT fwd+bwd (BASELINE configs[1], the headline); hier: hier-seq fwd+bwd "
                         "(configs[2]); train: hier-seq training step incl. fused clip+AdamW (configs[3]); meld: the "
                         "MELD-shaped training step on pooled (16, 768) features (configs[4])")
    ap.add_argument("--no-scaling-train", action="store_true",
                    help="N > 1: do not time the hierarchical-fusion training step after the headline region")
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--recast-weights", action="store_true",
                    help="re-cast all fp32 master weights to the bf16 shadow inside every step (round-1 behaviour; the "
                         "default casts only when a master changed: fwd+bwd leaves them unchanged, the fused AdamW of "
                         "the train workload writes the shadow itself)")
    ap.add_argument("--frozen-inputs", action="store_true",
                    help="inputs do not require grad (round-1 variant: no in-projection dgrads, no input-gradient sums)")
    ap.add_argument("--profile-steps", type=int, default=5)
    ap.add_argument("--dropout", type=float, default=0.0,
                    help="fusion_dropout (headline = 0, the parity-comparable setting; 0.1 = reference default)")
    ap.add_argument("--backend", choices=["nccl", "gloo"], default="nccl",
                    help="torch.distributed backend for N > 1 (nccl = RCCL over xGMI; gloo only to rehearse the "
                         "multi-rank code path on a box with fewer GPUs than ranks)")
    ap.add_argument("--shard-optimizer", action="store_true",
                    help="train workload, N > 1: ZeRO-1 optimiser step (reduce-scatter, AdamW on 1/N of the arena, all-gather "
                         "of the bf16 shadow) instead of all-reduce + replicated AdamW")
    ap.add_argument("--allreduce", choices=["bf16", "fp32"], default="bf16",
                    help="wire dtype of the gradient all-reduce for N > 1 (compute and accumulation stay as is)")
    args = ap.parse_args()

    if args.gpus < 1:
        raise SystemExit("bench.py: --gpus must be >= 1")
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # not under a launcher: become one (child process), before any GPU call in this process
        ndev = torch.cuda.device_count()
        if args.backend == "nccl" and ndev < args.gpus:
            raise SystemExit(f"bench.py: --gpus {args.gpus} but {ndev} GPU(s) visible on this node")
        raise SystemExit(self_launch(args.gpus, sys.argv[1:]))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the product path has no CPU fallback")
    ndev = torch.cuda.device_count()
    if local_rank >= ndev and args.backend == "nccl":
        raise SystemExit(f"rank {rank}: LOCAL_RANK {local_rank} but only {ndev} GPU(s) visible")
    dev_index = local_rank % max(ndev, 1)               # gloo rehearsal: ranks may share a GPU
    torch.cuda.set_device(dev_index)
    device = torch.device("cuda", dev_index)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world)

    if world > 1 and args.backend == "nccl":
        os.environ.setdefault("MMF_DP_STRICT", "1")     # a scaling run on RCCL must not silently measure a fallback schedule
    from mmfusion import arena as arena_mod, dp, synth
    if args.recast_weights:
        from models import fusion_layers as _flm
        _flm._RECAST = True
    if args.workload == "meld":
        cfg, model, xs = build_meld(device, rank, args.dropout if args.dropout > 0 else 0.1)
    else:
        cfg, model, xs = build(args.workload, device, rank, args.dropout, input_grads=not args.frozen_inputs)
    arena = arena_mod.ensure(model)
    use_graph = not args.no_graph
    compress = None if args.allreduce == "fp32" else "bf16"
    # N > 1: put the first half of the gradient arena on the wire while the second half of the deferred weight-gradient
    # launches is still computing.  The step is captured as three graphs sharing one memory pool — forward + backward
    # with the wgrad problems parked (ops.set_manual_wgrad_flush), then the two halves of the wgrad flush split by
    # gradient-arena offset — and the all-reduce of a finished arena range is started between the replays
    # (dp.allreduce_grads_range_async: RCCL's own stream, behind what the compute stream has enqueued so far).
    # MMF_DP_OVERLAP=0 keeps the one-graph step followed by the whole-arena all-reduce.
    want_overlap = world > 1 and use_graph and os.environ.get("MMF_DP_OVERLAP", "1") != "0"
    # number of wgrad parts / arena ranges (MMF_DP_PARTS, default 2): more parts hide more of the exchange under the
    # remaining wgrad launches and pay one more partly filled launch each; to be tuned against the scaling runs
    dp_parts = max(2, min(4, int(os.environ.get("MMF_DP_PARTS", "2"))))
    fallback = {"reason": None}

    def capture(fn):
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(3):
                fn()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            fn()
        return g

    def capture_split(body):
        """-> (graph of `body` with its wgrad problems parked, [graphs of the wgrad parts], [arena range bounds])"""
        from mmfusion import ops

        def eager_all():
            body()
            parts, _ = ops.split_wgrad_by_offset(ops.take_pending_wgrad(), dp_parts)
            for part in parts:
                ops.issue_wgrad(part)
        ops.set_manual_wgrad_flush(True)
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(3):
                eager_all()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        g1 = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g1):
            body()
        pend = ops.take_pending_wgrad()                 # operands live in g1's pool; referenced until the parts are captured
        parts, bounds = ops.split_wgrad_by_offset(pend, dp_parts)
        gparts = []
        for part in parts:
            g = None
            if part:
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g, pool=g1.pool()):
                    ops.issue_wgrad(part)
            gparts.append(g)
        del pend, parts
        ops.set_manual_wgrad_flush(False)
        return g1, gparts, bounds + [arena.numel]

    def try_capture_split(body):
        """capture_split, or None (with the automatic flush restored) if the three-graph capture fails on this
        software stack: the run then falls back to the one-graph step + one-shot all-reduce instead of aborting.
        The reason goes into the JSON line (config.overlap_fallback), not only to stderr."""
        from mmfusion import ops
        try:
            return capture_split(body)
        except Exception as e:                                   # noqa: BLE001 - any capture failure means: do not overlap
            if os.environ.get("MMF_DP_STRICT") == "1":            # a scaling run must not silently measure the fallback
                raise
            fallback["reason"] = f"{type(e).__name__}: {e}"[:300]
            if rank == 0:
                print(f"bench: overlapped exchange disabled ({fallback['reason']})", file=sys.stderr, flush=True)
            ops.set_manual_wgrad_flush(False)
            ops.take_pending_wgrad()
            torch.cuda.synchronize()
            return None

    def replay_split(g1, gparts, bounds):
        g1.replay()
        handles = []
        for i, g in enumerate(gparts):
            if g is not None:
                g.replay()
            handles.append(dp.allreduce_grads_range_async(arena, bounds[i], bounds[i + 1], compress=compress))
        for h in handles:
            h.finish()

    def make_runner(model, xs, arena, workload=None):
        """-> (run_step: one timed step incl. the exchange, profile_step: the same work WITHOUT any collective, overlap flag)"""
        workload = workload or args.workload
        overlap = want_overlap
        graph = graph2 = split = None
        if workload in ("train", "meld"):
            fwd_bwd, before_replay, exchange, opt_launch, sharded = make_train_step(model, xs, arena, world, args.allreduce,
                                                                                    rank, args.shard_optimizer)
            if sharded:
                overlap = False            # no gradient all-reduce to overlap: the sharded step reduce-scatters itself

            def eager_step():
                fwd_bwd()
                exchange()
                opt_launch()

            def profile_step():            # per-kernel timing pass: no collective
                fwd_bwd()
                opt_launch()
            if use_graph:                       # two graphs: the all-reduce sits between backward and AdamW
                # one eager step first: its AdamW launch has written the bf16 shadow (ParamArena.mark_shadow_fresh), so the
                # forward captured below holds no fp32->bf16 weight cast — every replayed step's shadow comes from the
                # optimiser kernel of the step before, as in FusionTrainStep
                eager_step()
                if overlap:
                    split = try_capture_split(fwd_bwd)
                    overlap = split is not None
                if not overlap:
                    graph = capture(fwd_bwd)
                if not sharded:            # the sharded step holds collectives: it runs eagerly between the replays
                    graph2 = capture(opt_launch)

            def run_step():
                if overlap:
                    replay_split(*split)
                    graph2.replay()
                elif graph is not None:
                    graph.replay()
                    exchange()
                    if graph2 is not None:
                        graph2.replay()
                    else:
                        opt_launch()
                else:
                    eager_step()
            return run_step, profile_step, overlap
        eager_step = make_step(workload, model, xs, arena)
        if overlap:
            split = try_capture_split(eager_step)
            overlap = split is not None
        if use_graph and not overlap:
            graph = capture(eager_step)

        def run_step():
            if overlap:
                replay_split(*split)
                return
            if graph is not None:
                graph.replay()
            else:
                eager_step()
            if world > 1:
                dp.allreduce_grads(arena, compress=compress)        # bucketed RCCL all-reduce (mean) of the flat gradient arena
        return run_step, eager_step, overlap

    run_step, profile_step, overlap = make_runner(model, xs, arena)
    if world > 1:                       # communicator set-up and the first collective's lazy work stay out of the timed region
        dist.all_reduce(torch.zeros(64, device=device))
        torch.cuda.synchronize()
    elapsed = timed_region(run_step, args.steps, args.warmup, world, dist, torch.cuda.synchronize,
                           device if args.backend == "nccl" else "cpu")

    # N > 1: the north_star's scaling target is the hierarchical-fusion TRAINING step (BASELINE configs[3]), not the headline MulT
    # step — time it here, inside the same process group, so that one SCALE run carries both numbers.  Every rank runs this.
    scaling_train = None
    if world > 1 and args.workload == "mult" and not args.no_scaling_train:
        _, model_t, xs_t = build("train", device, rank, args.dropout, input_grads=True)
        arena_t = arena_mod.ensure(model_t)
        fb_before = fallback["reason"]
        fallback["reason"] = None
        run_t, _, overlap_t = make_runner(model_t, xs_t, arena_t, "train")
        steps_t, warm_t = max(1, min(args.steps, 50)), max(1, min(args.warmup, 10))
        e_t = timed_region(run_t, steps_t, warm_t, world, dist, torch.cuda.synchronize,
                           device if args.backend == "nccl" else "cpu")
        scaling_train = {"metric": "hierarchical-fusion training step samples/sec at B=16/GPU d=768", "unit": "samples/s",
                         "value": round(world * 16 * steps_t / e_t, 2), "ms_per_step": round(e_t / steps_t * 1e3, 4),
                         "steps": steps_t, "warmup": warm_t, "n_gpus": world, "scaling": "weak", "parallelism": f"dp{world}",
                         "sharded_optimizer": bool(args.shard_optimizer), "allreduce_overlaps_wgrad": bool(overlap_t),
                         "overlap_fallback": fallback["reason"], "grad_allreduce": args.allreduce,
                         "strict": os.environ.get("MMF_DP_STRICT") == "1"}
        fallback["reason"] = fb_before
        del model_t, xs_t, arena_t, run_t

    # optional (MMF_BENCH_CHECKSUM=1): f64 |.|-sum and sum of the gradient arena after the last timed step, to compare
    # exchange schedules (overlapped vs one-shot all-reduce) on the same inputs
    checksum = None
    if os.environ.get("MMF_BENCH_CHECKSUM"):
        g64 = arena.grads.double()
        checksum = [float(g64.abs().sum()), float(g64.sum())]
    # ---- every rank leaves the group HERE; nothing below can wait for another rank ----
    leave_group(world, dist)
    if rank != 0:
        return
    S = synth.C2_SHAPES
    B = S["B"]
    ms_per_step = elapsed / args.steps * 1e3
    value = world * B * args.steps / elapsed

    # per-kernel HIP-event timing runs with the stream concurrency of the timed steps switched off (MulT's two
    # block groups, HierarchicalFusion's branch stream): a kernel that shares the chip with another stream's
    # kernel would be charged the other's time.  The dominant kernel (the deferred wgrad launch) runs after the
    # join either way, so its duration is the same in both modes (and in the rocprofv3 summary).
    from models import fusion_layers as _fl
    from mmfusion import ops as _ops
    saved_streams = (_fl._MULT_STREAMS, _fl._BRANCH_STREAM, _ops._WGRAD_EARLY)
    _fl._MULT_STREAMS, _fl._BRANCH_STREAM, _ops._WGRAD_EARLY = 1, False, False
    prof = kernel_profile(profile_step, args.profile_steps)
    _fl._MULT_STREAMS, _fl._BRANCH_STREAM, _ops._WGRAD_EARLY = saved_streams
    def roof(labels, name=None):
        """roofline object of one kernel label, or of several taken together (their FLOPs and times summed)"""
        ms = sum(prof[k]["ms_total"] for k in labels)
        fl = sum(prof[k]["flops_total"] for k in labels)
        n = sum(prof[k]["launches"] for k in labels)
        a = fl / (ms * 1e-3) / 1e12 if ms > 0 else 0.0
        r = {"bound": "mfma", "kernel": name or labels[0], "achieved": round(a, 2), "peak": BF16_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
             "frac": round(a / BF16_MFMA_PEAK_TFLOPS, 4), "us_per_step": round(ms * 1e3 / args.profile_steps, 1),
             "avg_launch_us": round(ms * 1e3 / max(1, n), 2), "launches_per_step": n // args.profile_steps}
        if len(labels) == 1 and args.workload == "mult":
            # the committed PMC passes are of the MulT workload: its per-launch byte counts do not describe the other workloads' launches
            r["traffic"], r["mfma_util_pmc"], r["traffic_and_util_source"] = pmc_lookup(labels[0])
        else:
            r["traffic"] = None
        return r
    # `roofline`: the kernel label with the largest time summed over its launches of a step (rounds 1-2's rule, restored in round 4;
    # round 3's line named the longest single launch instead — commit 71e2ace — which is the deferred wgrad launch: it runs alone
    # on the chip after the streams have joined, so its rocprofv3 duration in the timed command equals its stand-alone duration,
    # while the NT / NN launches share the chip with the other stream's kernels there and read longer than in this stand-alone pass).
    gemm_labels = sorted(k for k in prof if k.startswith("gemm"))
    dom = max(prof, key=lambda k: prof[k]["ms_total"])
    longest = max(prof, key=lambda k: prof[k]["ms_total"] / max(1, prof[k]["launches"]))
    roofline = roof([dom])
    roofline_longest = roof([longest])
    roofline_gemm_all = roof(gemm_labels, "all grouped GEMM launches of the step (" + ", ".join(gemm_labels) + ")") if gemm_labels else None
    kernels = {k: {"us_per_step": round(v["ms_total"] * 1e3 / args.profile_steps, 1),
                   "tflops": round(v["flops_total"] / (v["ms_total"] * 1e-3) / 1e12, 1) if v["ms_total"] > 0 else None,
                   "gflop_per_step": round(v["flops_total"] / args.profile_steps / 1e9, 1),
                   "launches_per_step": v["launches"] // args.profile_steps} for k, v in sorted(prof.items())}
    launched_flops = sum(v["flops_total"] for v in prof.values()) / args.profile_steps
    line = {
        "metric": "fusion fwd+bwd samples/sec at B=16 d=768", "value": round(value, 2), "unit": "samples/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 4),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "bf16",
        "data": "synthetic",
        "config": {"workload": ({"mult": "MulT fwd+bwd", "hier": "hier-seq fwd+bwd",
                                 "train": "hier-seq training step (fwd+bwd+clip+AdamW)"}[args.workload] +
                                f", B=16/GPU, T_text=512 T_audio=400 T_frames=30 d=768 H=8, fusion_dropout={args.dropout:g}")
                               if args.workload != "meld" else
                               "MELD-shaped training step (BASELINE configs[4]): 3 x (16, 768) features -> Linear(768, 512) -> "
                               "ModalityDropout(0.1) -> hierarchical fusion d=512 H=8 G=512 L=3 at T=1 -> classifier(7) -> "
                               "CE(ls 0.1) + 0.1 contrastive -> clip(1.0) + AdamW, B=16/GPU, dropout 0.1",
                   "input_gradients": not args.frozen_inputs,
                   "weight_shadow": "re-cast every step" if args.recast_weights else "cached while the masters are unchanged",
                   "global_batch": B * world, "parallelism": f"dp{world}",
                   "grad_allreduce": (args.allreduce if world > 1 else None), "allreduce_overlaps_wgrad": bool(overlap),
                   "sharded_optimizer": bool(args.shard_optimizer and world > 1),
                   "overlap_fallback": fallback["reason"], "graph_replay": bool(use_graph)},
        # algorithmic FLOPs of the GEMM / attention problems actually launched in one step / step time
        "step_gflop_launched": round(launched_flops / 1e9, 1),
        "step_tflops": round(launched_flops / (ms_per_step * 1e-3) / 1e12, 1),
        "roofline": roofline,
        "roofline_longest_launch": roofline_longest,
        "roofline_gemm_all": roofline_gemm_all,
        "kernels": kernels,
    }
    if scaling_train is not None:
        line["scaling_train"] = scaling_train
    if args.workload == "meld":
        meld_report(line, arena, ms_per_step, EXTRAS)
    af = prof.get("attn_fwd_kernel<96>")
    if af and af["ms_total"] > 0:
        # the north_star's named kernel: the MulT attention cores.  At these shapes the cores sit below the
        # 312 FLOP/B ridge (123-256 FLOP/B), so both roofs are reported: algorithmic FLOPs vs the bf16 MFMA peak
        # and algorithmic bytes (Q, K, V in, O out, once each) vs the HBM peak.
        sec = af["ms_total"] * 1e-3
        traffic, util, pmc_src = pmc_lookup("attn_fwd_kernel<96>")
        line["roofline_attention_fwd"] = {
            "kernel": "attn_fwd2n_kernel<96> (all nine MulT attention cores, 2 launches)",
            "mfma": {"achieved": round(af["flops_total"] / sec / 1e12, 1), "peak": BF16_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                     "frac": round(af["flops_total"] / sec / 1e12 / BF16_MFMA_PEAK_TFLOPS, 4)},
            "hbm": {"achieved": round(af["bytes_total"] / sec / 1e9, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": round(af["bytes_total"] / sec / 1e9 / HBM_PEAK_GBS, 4)},
            "traffic": traffic, "mfma_util_pmc": util, "traffic_and_util_source": pmc_src,
            "avg_launch_us": round(af["ms_total"] * 1e3 / af["launches"], 2)}
    if world == 1 and args.workload == "mult" and not args.frozen_inputs and not os.environ.get("MMF_BENCH_NO_FROZEN"):
        # the round-1 variant beside the headline: same step with inputs that do not require grad
        _, model2, xs2 = build(args.workload, device, rank, args.dropout, input_grads=False)
        arena2 = arena_mod.ensure(model2)
        run2, _, _ = make_runner(model2, xs2, arena2)
        e2 = timed_region(run2, args.steps, args.warmup, 1, None, torch.cuda.synchronize)
        line["frozen_inputs"] = {"value": round(B * args.steps / e2, 2), "unit": "samples/s",
                                 "ms_per_step": round(e2 / args.steps * 1e3, 4),
                                 "note": "inputs without requires_grad: no in-projection dgrads, no input-gradient sums"}
        del model2, xs2, arena2, run2
    if world == 1 and not args.no_cpu_baseline and args.workload == "mult":
        line["cpu_baseline"] = cpu_baseline(args.workload)
    if world == 1 and not args.no_cpu_baseline and args.workload == "meld":
        line["cpu_baseline"] = cpu_baseline_meld(model)
    if checksum is not None:
        line["grad_checksum"] = checksum
    print(json.dumps(line), flush=True)


if __name__ == "__main__":
    main()
