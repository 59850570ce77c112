"""One optimisation step of the path as a hipGraph (MI355X-first: HIP streams and graphs instead of a tracing compiler).

A ViT-B step is ~340 kernel launches from Python through ctypes; eagerly the host keeps up, but ~0.7-1 ms of the 42 ms
step were gaps between launches.  Captured once, the step is replayed with one `hipGraphLaunch`.  What changes from step
to step cannot live in kernel arguments (they are frozen at capture), so it lives in a 32-byte device record the library
reads at kernel entry (`vit_step_state_bind`, include/vit_amd.h): the dropout keys of the step and AdamW's step count /
bias corrections, advanced by the first node of the graph (`vit_step_advance`); the learning rate is written into that
record by the host whenever a scheduler changes it.

Scope: one process, one GPU (the RCCL exchange of N > 1 is not captured), FusedAdamW, no trainable input preprocessor, a
fixed batch shape.  The reference's step semantics are unchanged (zero_grad -> forward, dropout on -> backward -> clip the
global norm -> AdamW: src/basemodule.py:230-251); only the seed schedule of the dropout masks differs from the eager path
(masks are implementation-defined in the reference too).  The eager path stays the default everywhere but bench.py.

torch provides the capture plumbing (`torch.cuda.CUDAGraph` = hipStreamBeginCapture / hipGraphInstantiate / hipGraphLaunch
plus a private allocator pool so tensors created during capture keep their addresses).
"""
from __future__ import annotations

from typing import Optional

import torch

from . import _cabi
from . import functional as vf
from .optimizer import FusedAdamW

__all__ = ["GraphedTrainStep"]


class GraphedTrainStep:
    """`step(batch) -> loss` replaying a captured forward + backward + clip + AdamW over `module.model` (a MyViT).

    `batch` = (flux, error, labels); tensors whose storage differs from the captured ones are copied into the static input
    buffers first (a device-to-device copy of the batch)."""

    def __init__(self, module, optimizer: FusedAdamW, batch, warmup: int = 2):
        model = module.model
        eng = model.engine
        if not isinstance(optimizer, FusedAdamW):
            raise TypeError("GraphedTrainStep needs the FusedAdamW optimizer")
        if optimizer._extras or model.preprocessor is not None:
            raise ValueError("GraphedTrainStep: a trainable input preprocessor is outside the captured step")
        if getattr(module, "noise_level", 0):
            raise ValueError("GraphedTrainStep: on-the-fly noise draws a host seed per step; not captured")
        flux, _, labels = batch
        self.module, self.opt, self.eng = module, optimizer, eng
        dev = eng.flat.device
        eng._ensure_device_state()
        self.h = eng.handle()  # the engine's own handle: its calls read the bound per-step record
        # per-step device record: [key0, key1 (u32) | lr, bc1, rsqrt_bc2 (f32) | step (u32) | pad]
        self.state = torch.zeros(8, dtype=torch.int32, device=dev)
        self._lr = None
        self._set_lr(float(optimizer.param_groups[0]["lr"]))
        self.state[5] = int(optimizer._step)
        self.x = flux.detach().to(dev, torch.float32).contiguous().clone()
        self.labels = labels.detach().to(dev).contiguous().clone()
        # the copy into the static buffers is skipped only for the very tensor OBJECTS captured, unmodified since (holding the
        # references keeps their storage from being recycled for another batch at the same address)
        self._src = (flux, labels, flux._version, labels._version)
        self.dloss = torch.ones(1, dtype=torch.float32, device=dev)
        g = optimizer.param_groups[0]
        self._hyper = (g["betas"][0], g["betas"][1], g["eps"], g["weight_decay"])
        optimizer._ensure_state()
        module.train()
        # the warm-up steps below are real optimisation steps; parameters, moments and counters are put back afterwards so
        # that the first replay IS the run's next step
        eng._ensure_device_state()
        keep = (eng.flat.clone(), optimizer._m.clone(), optimizer._v.clone(), int(optimizer._step), int(eng.step_counter))
        # warm-up on a side stream (torch's capture protocol): sizes the arena / workspace, sets the kernels' LDS attributes
        # one stream inside the graph: two-branch graphs (the weight-gradient GEMMs on their second stream) replayed 0.7 ms per
        # step SLOWER than the same two streams launched eagerly on this stack.  The engine's own setting is put back after
        # the capture, so eager steps elsewhere (another batch shape, the bench's instrumented repetition) keep theirs.
        overlap_was = eng.overlap_dw
        eng.overlap_dw = False
        s = torch.cuda.Stream(device=dev)
        s.wait_stream(torch.cuda.current_stream(dev))
        _cabi.check(self.h.lib.vit_step_state_bind(self.h.h, self.state.data_ptr()), "vit_step_state_bind")
        try:
            with torch.cuda.stream(s):
                for _ in range(max(1, warmup)):
                    self._body()
            torch.cuda.current_stream(dev).wait_stream(s)
            self.graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.graph):
                self.loss = self._body()
            # The graph replays RAW POINTERS into the activation arena and the library workspace.  Hold the tensors: whatever
            # the engine or the handle do later (another batch size, a grown workspace), the captured addresses stay valid and
            # are nobody else's.  (The engine also keeps evaluation forwards in their own arena: engine._ensure_arena.)
            self._held = (eng.act, eng.tmp, self.h._ws)
        finally:
            # also on a failed warm-up / capture: the warm-up steps were real optimisation steps and must not leak into the run
            _cabi.check(self.h.lib.vit_step_state_bind(self.h.h, None), "vit_step_state_bind")
            eng.overlap_dw = overlap_was
            torch.cuda.synchronize(dev)
            eng.flat.copy_(keep[0]); optimizer._m.copy_(keep[1]); optimizer._v.copy_(keep[2])
            optimizer._step, eng.step_counter = keep[3], keep[4]
            self.state[5] = keep[3]
            self._dev_step = keep[3]  # host mirror of the record's step counter
            if eng.shadow is not None:
                vf.cast_f32_bf16(eng.flat, eng.shadow)
            eng.mark_shadow_fresh()
            torch.cuda.synchronize(dev)

    def _set_lr(self, lr: float):
        if lr != self._lr:
            self.state[2:3].view(torch.float32).fill_(lr)
            self._lr = lr

    def _body(self):
        """The captured sequence.  Host-side scalars below (seed, step) are frozen at capture; their per-step versions
        come from the bound device record.  Every call goes through the engine's handle (one workspace, held below)."""
        with vf.use_handle(self.h):
            return self._body_calls()

    def _body_calls(self):
        eng, opt = self.eng, self.opt
        lib, h = self.h.lib, self.h.h
        st = torch.cuda.current_stream(eng.flat.device).cuda_stream
        b1, b2, eps, wd = self._hyper
        _cabi.check(lib.vit_step_advance(h, eng.base_seed, b1, b2, st), "vit_step_advance")
        loss, _, _, _ = eng.forward(self.x, self.labels, training=True, need_grad=True)
        eng.backward(self.dloss)
        n = eng.layout.n_trainable
        sq = None
        if opt._clip is not None:
            sq = vf.grad_sqnorm(eng.grads[:n], out=opt._sq)
            opt.last_grad_norm = sq
        shadow = eng.shadow if eng.precision == "bf16" else None
        _cabi.check(lib.vit_adamw_step_dyn(h, eng.flat.data_ptr(), eng.grads.data_ptr(), opt._m.data_ptr(), opt._v.data_ptr(),
                                           None if shadow is None else shadow.data_ptr(), n, b1, b2, eps, wd,
                                           None if sq is None else sq.data_ptr(), float(opt._clip or 0.0), st),
                    "vit_adamw_step_dyn")
        return loss

    def step(self, batch) -> torch.Tensor:
        flux, _, labels = batch
        if flux is not self._src[0] or flux._version != self._src[2]:
            self.x.copy_(flux, non_blocking=True)
            self._src = (None, self._src[1], -1, self._src[3])
        if labels is not self._src[1] or labels._version != self._src[3]:
            self.labels.copy_(labels, non_blocking=True)
            self._src = (self._src[0], None, self._src[2], -1)
        self._set_lr(float(self.opt.param_groups[0]["lr"]))
        if self._dev_step != int(self.opt._step):
            # the record's step counter (AdamW bias corrections, dropout keys) is advanced by THIS graph's replays only: steps
            # taken elsewhere in between -- another captured shape (an epoch's partial last batch), eager steps -- put it back
            # in line with the optimizer's count before the replay reads it
            self.state[5:6].fill_(int(self.opt._step))
            self._dev_step = int(self.opt._step)
        # the bound record is only read by kernels of THIS graph; bind it around the replay so eager calls elsewhere (an
        # evaluation pass between steps) keep their host-seeded masks
        lib, h = self.h.lib, self.h.h
        _cabi.check(lib.vit_step_state_bind(h, self.state.data_ptr()), "vit_step_state_bind")
        self.graph.replay()
        _cabi.check(lib.vit_step_state_bind(h, None), "vit_step_state_bind")
        self.opt._step += 1          # host mirror of the device counter (no sync)
        self._dev_step += 1
        self.eng.step_counter += 1
        self.eng.mark_shadow_fresh()
        return self.loss

    __call__ = step
