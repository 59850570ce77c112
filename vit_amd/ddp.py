"""Data-parallel gradient exchange for the ViT step: the one collective the path has.

The reference gets this implicitly from Lightning's `strategy='ddp'` (src/hardware_utils.py:86-95): torch-DDP
all-reduces bucketed gradients during backward.  Here the gradients already live in ONE flat f32 buffer laid out in the
order backward completes them (tail, layer L-1 .. 0, embeddings), so each bucket is a contiguous slice: as soon as the
engine reports a slice complete, an asynchronous RCCL all-reduce (mean) is enqueued on it while the next layer's
backward kernels keep the compute stream busy; `finish()` joins before the norm-clip + AdamW kernel.  One process per
GPU, `torch.distributed` backend "nccl" (= RCCL over xGMI); "gloo" is supported for CPU rehearsals/tests.

The pooler parameters never receive a gradient (their output is unused, specvit.py:78), so they are simply outside every
bucket -- torch-DDP with find_unused_parameters=False would raise on them.
"""
from __future__ import annotations

from typing import List, Optional, Tuple

import torch
import torch.distributed as dist

__all__ = ["GradAllReducer", "ShardedGradReducer", "make_reducer", "broadcast_parameters", "init_distributed",
           "shard_indices", "exchange_active"]


def _single_rank_rehearsal() -> bool:
    """VIT_DIST_SINGLE=1: create the process group and run every collective of the step although WORLD_SIZE is 1.
    A one-GPU box cannot host two RCCL ranks (RCCL refuses duplicate devices), so this is how the RCCL calls themselves --
    in-place reduce-scatter / all-gather slices, async work handles, stream ordering against the second HIP stream -- are
    executed there (tests/test_ddp_gpu.py); the averaged values are trivially the rank's own."""
    import os

    return os.environ.get("VIT_DIST_SINGLE", "") not in ("", "0")


def exchange_active(group=None) -> bool:
    """True when the step has a gradient exchange to run: more than one rank, or the single-rank rehearsal."""
    if not dist.is_initialized():
        return False
    return dist.get_world_size(group) > 1 or _single_rank_rehearsal()


def init_distributed(backend: Optional[str] = None) -> Tuple[int, int, int]:
    """Read RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* (torchrun contract) and create the default process group."""
    import os

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if (world > 1 or _single_rank_rehearsal()) and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            # VIT_DIST_BACKEND=gloo lets several ranks share ONE GPU for rehearsals (RCCL refuses duplicate devices)
            backend = os.environ.get("VIT_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        if backend == "nccl":
            torch.cuda.set_device(local)
            dist.init_process_group(backend, rank=rank, world_size=world, device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    return rank, local, world


def broadcast_parameters(flat: torch.Tensor, src: int = 0, group=None):
    """DDP start-up semantics: every replica begins from rank 0's parameters."""
    if exchange_active(group):
        dist.broadcast(flat, src=src, group=group)


def shard_indices(n: int, rank: int, world: int, epoch: int, shuffle: bool, seed: int = 0) -> torch.Tensor:
    """DistributedSampler semantics (pad by wrapping so every rank gets ceil(n/world) samples)."""
    if shuffle:
        g = torch.Generator().manual_seed(seed + epoch)
        idx = torch.randperm(n, generator=g)
    else:
        idx = torch.arange(n)
    per = -(-n // world)
    total = per * world
    if total > n:
        idx = torch.cat([idx, idx[: total - n]])
    return idx[rank:total:world]


class GradAllReducer:
    """Mean all-reduce of the slices [lo, hi) of one flat gradient buffer, launched as they become ready."""

    def __init__(self, grads_getter, buckets: List[Tuple[int, int]], group=None, max_bucket_elems: int = 64 << 20,
                 grad_dtype: str = "fp32"):
        """`grad_dtype='bf16'` (`train.ddp_grad_dtype`, SURVEY.md section 5: 172 MB instead of 343 MB per step on the links):
        each bucket is cast into a bf16 send buffer, that buffer is all-reduced, and the result is cast back into the fp32
        gradient buffer at the join.  The SUM then runs in bf16 inside the collective library (RCCL has no fp32-accumulate
        mode for bf16 payloads): every rank's gradient is rounded to 8 significant bits before it is added, so this is an
        option with a measured parity cost (tests/test_ddp_cpu.py, tests/test_ddp_gpu.py), never the default -- the
        reference's DDP exchanges fp32 gradients (src/hardware_utils.py:86-95)."""
        self._get = grads_getter
        self.buckets = buckets
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.active = exchange_active(group)
        if max_bucket_elems % 8:
            raise ValueError("max_bucket_elems must be a multiple of 8 (16-byte aligned bf16 pieces)")
        self.max_bucket_elems = max_bucket_elems
        gd = str(grad_dtype or "fp32").lower()
        if gd not in ("fp32", "f32", "float32", "32", "bf16", "bfloat16"):
            raise ValueError(f"unknown train.ddp_grad_dtype '{grad_dtype}' (use 'fp32' or 'bf16')")
        self.grad_dtype = "bf16" if gd in ("bf16", "bfloat16") else "fp32"
        self._send: Optional[torch.Tensor] = None
        self._pending = []
        self._seen = 0
        self.bytes_reduced = 0
        self.mode = "allreduce"
        # what one step puts on the wire (for bench.py's `comm` object)
        self.calls_per_step = sum(-(-(hi - lo) // max_bucket_elems) for lo, hi in buckets if hi > lo)
        self.bytes_per_step = (2 if self.grad_dtype == "bf16" else 4) * sum(hi - lo for lo, hi in buckets if hi > lo)

    @staticmethod
    def _to_bf16(src: torch.Tensor, dst: torch.Tensor):
        if src.is_cuda:
            from . import functional as vf

            vf.cast_f32_bf16(src, dst)
        else:
            dst.copy_(src)

    @staticmethod
    def _from_bf16(src: torch.Tensor, dst: torch.Tensor, scale: float):
        if src.is_cuda:
            from . import functional as vf

            vf.cast_bf16_f32(src, dst, scale)
        else:
            dst.copy_(src.float() * scale)

    def bucket_ready(self, lo: int, hi: int):
        """Engine callback (called on the host right after the kernels that complete grads[lo:hi] were enqueued)."""
        if not self.active or hi <= lo:
            return
        g = self._get()
        backend = dist.get_backend(self.group)
        if self.grad_dtype == "bf16" and (self._send is None or self._send.numel() != g.numel() or self._send.device != g.device):
            self._send = torch.empty(g.numel(), dtype=torch.bfloat16, device=g.device)
        # an all-reduce bigger than max_bucket_elems is split so that the first pieces are on the wire early
        for a in range(lo, hi, self.max_bucket_elems):
            b = min(hi, a + self.max_bucket_elems)
            t, back = g[a:b], None
            if self.grad_dtype == "bf16":
                back, t = t, self._send[a:b]
                self._to_bf16(back, t)
            if backend == "nccl":
                w = dist.all_reduce(t, op=dist.ReduceOp.AVG, group=self.group, async_op=True)
                self._pending.append((w, t, back, 1.0))
            else:
                if t.is_cuda:  # gloo reads the tensor on the host side: the producing kernels must have finished
                    torch.cuda.current_stream(t.device).synchronize()
                w = dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
                # fp32 path: scale None = DIVIDE by the world size at the join, as torch-DDP does (x / 3 and x * (1/3) differ
                # in the last bit); the multiplicative scale exists only for the bf16 cast-back kernel
                self._pending.append((w, t, back, (1.0 / self.world) if back is not None else None))
            self.bytes_reduced += t.numel() * t.element_size()
        self._seen += 1

    def finish(self):
        """Join every outstanding all-reduce (the compute stream waits on RCCL's stream; no host sync with nccl)."""
        for w, t, back, scale in self._pending:
            w.wait()
            if back is not None:  # bf16 exchange: back into the fp32 gradient buffer (and the mean, where the library summed)
                self._from_bf16(t, back, scale)
            elif scale is None:
                t.div_(self.world)
        self._pending = []
        self._seen = 0


class ShardedGradReducer(GradAllReducer):
    """SURVEY.md section 8e's second schedule ("zero1"): reduce-scatter the gradient buckets, run AdamW on the 1/world
    shard each rank owns (1/world of the optimizer's 30 B/parameter HBM traffic), all-gather the updated fp32 parameters
    (the bf16 shadow is re-cast locally).  Same bytes on the xGMI links as the all-reduce (reduce-scatter + all-gather IS a
    ring all-reduce), but the all-gather half sits after the optimizer instead of overlapping backward, so this pays when the
    optimizer pass is the larger cost; it is a switch (`train.ddp_exchange: zero1` / VIT_DDP_EXCHANGE), not the default.

    A bucket is sharded when its length divides by 8 * world (shard starts stay 32-byte aligned); other buckets (the few
    thousand tail / embedding elements of small configs) are all-reduced and updated redundantly on every rank.
    The clipping norm = sum over the all-reduced buckets (identical everywhere) + all-reduce of the ranks' shard sums.
    gloo has no reduce-scatter: there the bucket is all-reduced and the rank then simply uses its shard (same values)."""

    def __init__(self, grads_getter, buckets, group=None, max_bucket_elems: int = 64 << 20, grad_dtype: str = "fp32"):
        super().__init__(grads_getter, buckets, group, max_bucket_elems, grad_dtype)
        if self.grad_dtype != "fp32":
            raise ValueError("train.ddp_grad_dtype='bf16' is offered with the all-reduce exchange only (the sharded schedule "
                             "reduce-scatters in place in the fp32 gradient buffer)")
        self.mode = "zero1"
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.sharded = [(hi - lo) > 0 and (hi - lo) % (8 * self.world) == 0 for lo, hi in buckets]
        self._sharded_of = {(lo, hi): sh for (lo, hi), sh in zip(buckets, self.sharded)}
        self.calls_per_step = sum(1 for lo, hi in buckets if hi > lo)
        self.gather_calls_per_step = sum(self.sharded)

    def shard(self, lo: int, hi: int) -> Tuple[int, int]:
        c = (hi - lo) // self.world
        return lo + self.rank * c, lo + (self.rank + 1) * c

    def is_sharded(self, lo: int, hi: int) -> bool:
        return self.active and self._sharded_of[(lo, hi)]

    def bucket_ready(self, lo: int, hi: int):
        if not self.active or hi <= lo:
            return
        if not self.is_sharded(lo, hi) or dist.get_backend(self.group) != "nccl":
            return super().bucket_ready(lo, hi)
        g = self._get()
        a, b = self.shard(lo, hi)
        w = dist.reduce_scatter_tensor(g[a:b], g[lo:hi], op=dist.ReduceOp.AVG, group=self.group, async_op=True)
        self._pending.append((w, g[a:b], None, 1.0))
        self.bytes_reduced += (hi - lo) * 4

    def all_gather_params(self, flat: torch.Tensor):
        """After the sharded update: every rank receives the other ranks' updated slices of the sharded buckets."""
        if not self.active:
            return
        works = []
        for (lo, hi), sh in zip(self.buckets, self.sharded):
            if not sh:
                continue
            a, b = self.shard(lo, hi)
            if dist.get_backend(self.group) == "nccl":
                works.append(dist.all_gather_into_tensor(flat[lo:hi], flat[a:b], group=self.group, async_op=True))
            else:
                if flat.is_cuda:
                    torch.cuda.current_stream(flat.device).synchronize()
                c = b - a
                works.append(dist.all_gather([flat[lo + r * c:lo + (r + 1) * c] for r in range(self.world)], flat[a:b].clone(),
                                             group=self.group, async_op=True))
        for w in works:
            w.wait()


def make_reducer(kind: str, engine, group=None, grad_dtype: str = "fp32", max_bucket_elems: int = 64 << 20):
    """'allreduce' (default) or 'zero1' over the engine's flat gradient buffer and its backward-ordered buckets."""
    kind = (kind or "allreduce").lower()
    if kind in ("allreduce", "all_reduce", "ar"):
        return GradAllReducer(lambda: engine.grads, engine.layout.buckets(), group, max_bucket_elems, grad_dtype)
    if kind in ("zero1", "reduce_scatter", "rs"):
        return ShardedGradReducer(lambda: engine.grads, engine.layout.buckets(), group, max_bucket_elems, grad_dtype)
    raise ValueError(f"unknown train.ddp_exchange '{kind}' (use 'allreduce' or 'zero1')")
