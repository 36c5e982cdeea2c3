"""Data-parallel exchange for the fusion path: one process per GPU, parameters replicated, batch
sharded; the only collective is the mean all-reduce of the flat fp32 gradient arena
(``ParamArena.grads``) — RCCL over xGMI when the process group's backend is ``nccl``, gloo on CPU
in the tests.  The reference has no distributed code at all; this is the
MI355X-native addition of SURVEY.md section 8(e).

xGMI is point-to-point (7 links x ~153 GB/s per GPU), so ring all-reduce is per-link bound:
few, large buckets (default 64 MiB) keep every link streaming and still let the first buckets
start while the rest of backward runs (``allreduce_flat_async`` + ``wait``).
"""
from __future__ import annotations

from typing import List, Optional

import torch
import torch.distributed as dist

DEFAULT_BUCKET_BYTES = 64 << 20


def bucket_bounds(numel: int, elem_size: int, bucket_bytes: int = DEFAULT_BUCKET_BYTES) -> List[tuple]:
    """[(start, end)) element ranges covering [0, numel), each at most bucket_bytes, 64-element aligned."""
    per = max(64, (bucket_bytes // elem_size) // 64 * 64)
    return [(s, min(numel, s + per)) for s in range(0, numel, per)]


class _Pending:
    def __init__(self, flat, works, world, average):
        self.flat, self.works, self.world, self.average = flat, works, world, average

    def wait(self) -> None:
        for w in self.works:
            w.wait()
        if self.average and self.world > 1:
            self.flat.mul_(1.0 / self.world)


def allreduce_flat_async(flat: torch.Tensor, group=None, average: bool = True,
                         bucket_bytes: int = DEFAULT_BUCKET_BYTES) -> _Pending:
    """Start bucketed SUM all-reduces of a flat contiguous tensor; ``.wait()`` finishes and averages."""
    if flat.dim() != 1 or not flat.is_contiguous():
        raise ValueError("allreduce_flat expects a flat contiguous tensor (the gradient arena)")
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    works = []
    if world > 1:
        for s, e in bucket_bounds(flat.numel(), flat.element_size(), bucket_bytes):
            works.append(dist.all_reduce(flat[s:e], op=dist.ReduceOp.SUM, group=group, async_op=True))
    return _Pending(flat, works, world, average)


def allreduce_flat(flat: torch.Tensor, group=None, average: bool = True,
                   bucket_bytes: int = DEFAULT_BUCKET_BYTES) -> None:
    allreduce_flat_async(flat, group, average, bucket_bytes).wait()


def allreduce_grads(arena, group=None, bucket_bytes: int = DEFAULT_BUCKET_BYTES,
                    compress: Optional[str] = None) -> None:
    """Mean all-reduce of every parameter gradient of a ``mmfusion.arena.ParamArena`` in place.

    ``compress="bf16"`` sends bf16 over the wire: grads (fp32) -> bf16 wire buffer (the HIP cast kernel),
    bucketed SUM all-reduce in bf16, -> fp32, * 1/world.  Halves the bytes on xGMI (103 MB instead of
    206 MB for MulT at d=768); each rank's contribution is rounded to 8 significant bits once and the
    ring sums in bf16, the usual trade of bf16 gradient compression.  Default (None) is exact fp32."""
    if compress is None:
        allreduce_flat(arena.grads, group, True, bucket_bytes)
        return
    if compress != "bf16":
        raise ValueError("compress must be None or 'bf16'")
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    if world == 1:
        return
    wire = getattr(arena, "_wire_bf16", None)
    if wire is None or wire.numel() != arena.grads.numel():
        wire = torch.empty(arena.grads.numel(), dtype=torch.bfloat16, device=arena.grads.device)
        arena._wire_bf16 = wire
    # Pipelined per bucket: narrow(b) -> all-reduce(b) [async, RCCL's stream] -> widen(b).  The narrowing of
    # bucket b+1 runs on the compute stream while bucket b is on the wire, and bucket b is widened (and averaged,
    # same kernel) while bucket b+1 is on the wire; only the first narrowing and the last widening are exposed.
    gpu = arena.grads.is_cuda
    if gpu:
        from . import lib
        L = lib.load()
    bounds = bucket_bounds(wire.numel(), wire.element_size(), bucket_bytes)
    pending = []
    for s, e in bounds:
        if gpu:
            lib.check(L.mmf_cast_f32_to_bf16(arena.grads.data_ptr() + 4 * s, wire.data_ptr() + 2 * s, e - s, lib.stream_ptr()))
        else:                               # CPU (gloo tests): same arithmetic with torch casts
            wire[s:e].copy_(arena.grads[s:e])
        pending.append((s, e, dist.all_reduce(wire[s:e], op=dist.ReduceOp.SUM, group=group, async_op=True)))
    for s, e, work in pending:
        work.wait()
        if gpu:
            lib.check(L.mmf_cast_bf16_to_f32_scaled(wire.data_ptr() + 2 * s, arena.grads.data_ptr() + 4 * s, e - s,
                                                    1.0 / world, lib.stream_ptr()))     # widen and average in one pass
        else:
            arena.grads[s:e].copy_(wire[s:e])
            arena.grads[s:e].mul_(1.0 / world)


class _RangePending:
    """Handle of ``allreduce_grads_range_async``: ``finish()`` waits for the range's buckets and writes the averaged
    gradients back (widen + 1/world in one kernel for the bf16 wire format)."""

    def __init__(self, arena, items, world, compress):
        self.arena, self.items, self.world, self.compress = arena, items, world, compress

    def finish(self) -> None:
        a = self.arena
        for s, e, work in self.items:
            work.wait()
            if self.compress == "bf16":
                if a.grads.is_cuda:
                    from . import lib
                    lib.check(lib.load().mmf_cast_bf16_to_f32_scaled(a._wire_bf16.data_ptr() + 2 * s, a.grads.data_ptr() + 4 * s,
                                                                     e - s, 1.0 / self.world, lib.stream_ptr()))
                else:
                    a.grads[s:e].copy_(a._wire_bf16[s:e])
                    a.grads[s:e].mul_(1.0 / self.world)
            else:
                a.grads[s:e].mul_(1.0 / self.world)
        self.items = []


def allreduce_grads_range_async(arena, start: int, end: int, group=None, bucket_bytes: int = DEFAULT_BUCKET_BYTES,
                                compress: Optional[str] = None) -> _RangePending:
    """Start the mean all-reduce of gradient-arena elements [start, end) and return at once: the collective runs on
    the backend's own stream behind everything enqueued on the current stream so far, so kernels enqueued AFTER this
    call (the rest of the weight-gradient launches) run beside it.  Used to put the first arena range on the wire
    while the second half of the deferred wgrad flush is still computing (bench.py, N > 1).  ``finish()`` on the
    returned handle completes it; ranges must not overlap and nothing may write [start, end) until then."""
    if compress not in (None, "bf16"):
        raise ValueError("compress must be None or 'bf16'")
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    if world == 1 or end <= start:
        return _RangePending(arena, [], world, compress)
    items = []
    if compress == "bf16":
        wire = getattr(arena, "_wire_bf16", None)
        if wire is None or wire.numel() != arena.grads.numel():
            wire = torch.empty(arena.grads.numel(), dtype=torch.bfloat16, device=arena.grads.device)
            arena._wire_bf16 = wire
        gpu = arena.grads.is_cuda
        if gpu:
            from . import lib
            L = lib.load()
        for s, e in bucket_bounds(end - start, 2, bucket_bytes):
            s, e = s + start, e + start
            if gpu:
                lib.check(L.mmf_cast_f32_to_bf16(arena.grads.data_ptr() + 4 * s, wire.data_ptr() + 2 * s, e - s, lib.stream_ptr()))
            else:
                wire[s:e].copy_(arena.grads[s:e])
            items.append((s, e, dist.all_reduce(wire[s:e], op=dist.ReduceOp.SUM, group=group, async_op=True)))
    else:
        for s, e in bucket_bounds(end - start, 4, bucket_bytes):
            s, e = s + start, e + start
            items.append((s, e, dist.all_reduce(arena.grads[s:e], op=dist.ReduceOp.SUM, group=group, async_op=True)))
    return _RangePending(arena, items, world, compress)


class BackwardExchange:
    """The gradient exchange moved INTO backward (SURVEY.md 8e as specified: buckets in reverse-autograd order, each put
    on the wire as soon as backward has produced it).  ``install()`` registers a round hook with the deferred wgrad
    queue (``ops.set_wgrad_rounds``): every round's weight-gradient GEMMs are issued at once on the wgrad side stream and
    the mean all-reduce of the gradient-arena runs they complete is started right behind them, while the main stream
    goes on with the backward pass; ``finish()`` (after ``arena.finalize_grads()``) starts the exchange of everything no
    round covered — biases, LayerNorm vectors, torch-produced gradients, regions that got no gradient — and waits for
    all of it, so the arena holds the mean over ranks of every element.

    **Every rank issues the same collectives in the same order** (round 4).  What goes on the wire early is a PLAN —
    per round, the element runs of the arena — that all ranks hold identically: a step records the runs its own rounds
    completed, ``finish()`` compares a hash of that record across ranks (one three-number MAX all-reduce) and only an
    agreed record becomes the next step's plan.  While a plan is active a round is sent early only if it completes exactly the
    planned runs in the planned position; the first deviation (a branch switched off by modality dropout on this rank only, a
    weight used a different number of times, more or fewer problems than last step) stops this rank's early sends, and
    ``finish()`` sends the remaining planned rounds — in plan order — before the gaps, so the sequence of collectives is the
    plan's on every rank whatever its own backward did.  A gradient region written AGAIN after its round went out (a late
    writer the plan did not know) first waits for that round's collective, accumulates onto the mean, and is marked dirty;
    dirty flags are part of the three-number all-reduce and any rank's flag makes every rank re-send the planned runs: the mean
    of (mean + late contribution) is the correct mean, because the mean of values that are already identical is the value.

    Eager steps only: nothing here is captured into a hipGraph (the collectives run on RCCL's own stream).  Unmeasured
    on RCCL hardware; the logic is covered by tests/test_dp_rounds_cpu.py on two gloo ranks, including a
    step whose graph differs between the ranks."""

    def __init__(self, arena, rounds: int = 4, compress: Optional[str] = "bf16", group=None,
                 bucket_bytes: int = DEFAULT_BUCKET_BYTES):
        self.arena, self.rounds, self.compress, self.group, self.bucket_bytes = arena, rounds, compress, group, bucket_bytes
        self._handles: List[tuple] = []                  # (runs, [_RangePending]) per early-sent round, in order
        self._plan: Optional[List[List[tuple]]] = None   # agreed by all ranks at the end of the previous step
        self._observed: List[List[tuple]] = []           # this step's rounds, as run lists
        self._sent_rounds = 0                            # planned rounds this rank has put on the wire so far
        self._deviated = False
        self._dirty = False
        self.rounds_seen = 0                             # hook calls of the current step (tests, logging)
        self.resends = 0                                 # steps in which the planned runs were exchanged a second time (tests)

    def install(self) -> "BackwardExchange":
        from . import ops
        ops.set_wgrad_rounds(self.rounds, self._on_round)
        return self

    def remove(self) -> None:
        from . import ops
        ops.set_wgrad_rounds(0, None)

    # -- overridable: how a round's GEMMs are issued (the CPU logic test substitutes torch arithmetic) -----------------
    def _issue(self, problems) -> None:
        from . import ops
        if self.arena.grads.is_cuda:
            ops._issue_wgrad_side(problems, all_streams=True)      # on the wgrad stream, behind every producer stream
        else:
            ops.issue_wgrad(problems)

    def _stream(self):
        from . import ops
        if self.arena.grads.is_cuda and ops._wgrad_stream is not None:
            return torch.cuda.stream(ops._wgrad_stream)
        import contextlib
        return contextlib.nullcontext()

    def _span(self, t: torch.Tensor) -> tuple:
        """(first, one past the last) arena element a gradient view may be written at: for a row-strided view (a column block of
        a packed weight) the last row ends at (rows - 1) * ld + cols, not at numel (as ops._wgrad_rounds)"""
        first = (t.data_ptr() - self.arena.grads.data_ptr()) // 4
        n = (t.shape[0] - 1) * t.stride(0) + t.shape[1] if t.dim() == 2 else t.numel()
        return first, first + n

    def _runs(self, problems) -> List[tuple]:
        """coalesced (start, end) element runs of the weight-gradient regions of `problems` inside the arena"""
        spans = sorted(self._span(q[2]) for q in problems)
        runs: List[list] = []
        for s0, e0 in spans:
            s0, e0 = s0 // 64 * 64, (e0 + 63) // 64 * 64            # parameters start on 64-element boundaries
            if runs and s0 <= runs[-1][1]:
                runs[-1][1] = max(runs[-1][1], e0)
            else:
                runs.append([s0, e0])
        return [(a, min(b, self.arena.numel)) for a, b in runs]

    def _send(self, runs) -> None:
        with self._stream():
            hs = [allreduce_grads_range_async(self.arena, s0, e0, self.group, self.bucket_bytes, self.compress) for s0, e0 in runs]
        self._handles.append((runs, hs))

    def _late_writers(self, problems) -> None:
        """a problem about to write a region whose round is already on the wire: wait for that collective first (the GEMM
        must not run under it), then the accumulate lands on the mean and the region is exchanged again in finish()"""
        for q in problems:
            s0, e0 = self._span(q[2])
            for runs, hs in self._handles:
                if any(a < e0 and s0 < b for a, b in runs):
                    for h in hs:
                        h.finish()
                    self._dirty = True

    def _on_round(self, problems, final: bool) -> None:
        self.rounds_seen += 1
        if problems:
            self._late_writers(problems)
            self._issue(problems)
        if final:
            return                                       # the last round's regions go out with the rest in finish()
        runs = self._runs(problems) if problems else []
        i = len(self._observed)
        self._observed.append(runs)
        if self._plan is None or self._deviated:
            return
        if i < len(self._plan) and runs == self._plan[i] and self._sent_rounds == i:
            self._send(runs)
            self._sent_rounds = i + 1
        else:
            self._deviated = True                        # this rank's backward left the plan: no more early sends this step

    def _agree(self, digest: int) -> tuple:
        """-> (any rank dirty, all ranks hold the same digest): ONE MAX all-reduce of [dirty, digest, -digest]"""
        world = dist.get_world_size(self.group) if dist.is_initialized() else 1
        if world == 1:
            return self._dirty, True
        dev = self.arena.grads.device
        t = torch.tensor([int(self._dirty), digest, -digest], dtype=torch.int64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX, group=self.group)
        v = t.tolist()
        return bool(v[0]), v[1] == -v[2] and v[1] == digest

    def finish(self) -> None:
        """Call after backward AND ``arena.finalize_grads()``: exchange what no round sent, then wait for everything."""
        if self.arena.grads.is_cuda:
            from . import ops
            if ops._wgrad_stream is not None:            # the final round's GEMMs ran on the wgrad stream
                torch.cuda.current_stream().wait_stream(ops._wgrad_stream)
        plan = self._plan or []
        for i in range(self._sent_rounds, len(plan)):    # planned rounds this rank did not reach: same collectives, only later
            self._send(plan[i])
        covered = sorted(r for runs in plan for r in runs)
        pos, rest = 0, []
        for s0, e0 in covered + [(self.arena.numel, self.arena.numel)]:
            if s0 > pos:
                rest.append((pos, s0))
            pos = max(pos, e0)
        tail = [allreduce_grads_range_async(self.arena, a, b, self.group, self.bucket_bytes, self.compress) for a, b in rest]
        for _, hs in self._handles:
            for h in hs:
                h.finish()
        for h in tail:
            h.finish()
        # one small all-reduce: does any rank hold a region written after its round went out, and do all ranks agree on what
        # this step's rounds completed (the candidate plan of the next step)?
        import hashlib
        rec = self._observed if not self._deviated or self._plan is None else None
        digest = int.from_bytes(hashlib.sha256(repr(rec).encode()).digest()[:7], "big") if rec is not None else 0
        dirty, same = self._agree(digest)
        if dirty:
            self.resends += 1
            again = [allreduce_grads_range_async(self.arena, a, b, self.group, self.bucket_bytes, self.compress) for a, b in covered]
            for h in again:
                h.finish()
        self._plan = ([r for r in rec if r] or None) if (same and rec is not None and digest != 0) else None
        self._handles, self._observed, self._sent_rounds, self._deviated, self._dirty, self.rounds_seen = [], [], 0, False, False, 0


def broadcast_params(arena, src: int = 0, group=None) -> None:
    """Make the replicas identical (rank ``src``'s masters win), then refresh the bf16 shadow."""
    if hasattr(arena, "require_fresh_masters"):
        arena.require_fresh_masters("broadcasting the fp32 masters")
    if dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.broadcast(arena.master, src=src, group=group)
    arena.refresh(force=True)


def shard_batch(x: torch.Tensor, rank: int, world: int) -> torch.Tensor:
    """Rows [rank*B/world, (rank+1)*B/world) of a batch-first tensor (B must divide evenly)."""
    B = x.shape[0]
    if B % world:
        raise ValueError(f"global batch {B} is not divisible by world size {world}")
    per = B // world
    return x[rank * per:(rank + 1) * per]
