"""Adam over one flat parameter buffer, one kernel launch per step.

``torch.optim.Adam`` (``rwgnn2/main.py:79``) issues ~45 small launches for the 10 parameter tensors of
``RW_NN``; at ~0.5 ms per training step that is a fifth of the step.  Here parameters, gradients and
both moment estimates are contiguous fp32 buffers and the update is ``rwk_adam_step``.  Same arithmetic
as ``torch.optim.Adam`` (no amsgrad, no maximize); the step count and the learning rate live on the
device, so the update can be captured in a HIP graph and a scheduler can change the rate between replays.
"""
from __future__ import annotations

import torch

from . import _lib
from .dp import FlatGradBucket

__all__ = ["FlatAdam"]


class FlatAdam:
    def __init__(self, module: torch.nn.Module, bucket: FlatGradBucket = None, lr=1e-3, betas=(0.9, 0.999),
                 eps=1e-8, weight_decay=0.0):
        self.bucket = bucket if bucket is not None else FlatGradBucket(module)
        params = self.bucket.params
        dev = params[0].device
        if dev.type != "cuda":
            raise RuntimeError("FlatAdam runs on HIP devices only")
        self.flat_param = torch.cat([p.detach().reshape(-1) for p in params]).contiguous()
        off = 0
        for p in params:                      # parameters become views of the flat buffer
            n = p.numel()
            p.data = self.flat_param[off:off + n].view_as(p)
            off += n
        self.exp_avg = torch.zeros_like(self.flat_param)
        self.exp_avg_sq = torch.zeros_like(self.flat_param)
        self.lr = torch.tensor([float(lr)], dtype=torch.float32, device=dev)
        self.betas = (float(betas[0]), float(betas[1]))
        self.eps = float(eps)
        self.weight_decay = float(weight_decay)
        lib = _lib.load()
        self.step_counters = torch.zeros(lib.rwk_adam_n_counters(self.flat_param.numel()), dtype=torch.int64,
                                         device=dev)

    def set_lr(self, lr: float):
        self.lr.fill_(float(lr))

    def zero_grad(self):
        self.bucket.zero()

    @torch.no_grad()
    def step(self, gather: bool = False):
        """One Adam update from ``bucket.flat``.  ``gather=True``: take the gradients from the parameters'
        ``.grad`` tensors instead (after ``bucket.release()`` and a backward pass, without ``bucket.gather()``)
        and fill ``bucket.flat`` on the way: one launch for a single-process step."""
        lib = _lib.load()
        dev = self.flat_param.device
        params = self.bucket.params
        if gather and len(params) > 16:
            self.bucket.gather()
            gather = False
        if gather:
            import ctypes
            grads = [p.grad for p in params]
            grads = [g if g is None or (g.dtype is torch.float32 and g.is_contiguous())
                     else g.detach().contiguous().float() for g in grads]
            n = len(grads)
            srcs = (ctypes.c_void_p * n)(*[None if g is None else g.data_ptr() for g in grads])
            sizes = (ctypes.c_int64 * n)(*[p.numel() for p in params])
            with _lib.device_guard(dev):
                _lib.check(lib.rwk_adam_step_chunks(
                    self.flat_param.data_ptr(), srcs, sizes, n, self.bucket.flat.data_ptr(), self.exp_avg.data_ptr(),
                    self.exp_avg_sq.data_ptr(), self.flat_param.numel(), self.lr.data_ptr(), self.betas[0],
                    self.betas[1], self.eps, self.weight_decay, self.step_counters.data_ptr(),
                    _lib.current_stream(dev)), "rwk_adam_step_chunks")
            self._keep = grads                     # the launch reads them asynchronously
            return
        _lib.check(lib.rwk_adam_step(self.flat_param.data_ptr(), self.bucket.flat.data_ptr(),
                                     self.exp_avg.data_ptr(), self.exp_avg_sq.data_ptr(), self.flat_param.numel(),
                                     self.lr.data_ptr(), self.betas[0], self.betas[1], self.eps, self.weight_decay,
                                     self.step_counters.data_ptr(), _lib.current_stream(dev)), "rwk_adam_step")
