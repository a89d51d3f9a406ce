"""``F.cross_entropy(output, y)`` (``rwgnn2/main.py:85,92``: mean reduction, default arguments) in one launch each
way instead of torch's six (log-softmax, nll-loss, two fills and their backward kernels cost more than the
model's own forward pass on a 1024-graph batch)."""
from __future__ import annotations

import torch

from . import _lib

__all__ = ["cross_entropy"]


class _CrossEntropy(torch.autograd.Function):
    @staticmethod
    def forward(ctx, scores, labels):
        lib = _lib.load()
        if not scores.is_cuda:
            raise RuntimeError("gnn_amd runs on HIP devices only (no CPU fallback)")
        if scores.dim() != 2 or labels.dim() != 1 or labels.shape[0] != scores.shape[0]:
            raise ValueError("expected scores (G, C) and labels (G,)")
        if labels.dtype != torch.int64:
            raise TypeError("labels must be int64")
        g, nc = scores.shape
        dev = scores.device
        with _lib.device_guard(dev):
            x = scores.detach().contiguous().float()
            y = labels.contiguous()
            loss = torch.empty((), dtype=torch.float32, device=dev)
            _lib.check(lib.rwk_cross_entropy_forward(x.data_ptr(), y.data_ptr(), g, nc, loss.data_ptr(),
                                                     _lib.current_stream(dev)), "rwk_cross_entropy_forward")
        ctx.save_for_backward(x, y)
        return loss

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, d_loss):
        lib = _lib.load()
        x, y = ctx.saved_tensors
        g, nc = x.shape
        dev = x.device
        with _lib.device_guard(dev):
            d_loss = d_loss.contiguous().float()
            d_x = torch.empty_like(x)
            _lib.check(lib.rwk_cross_entropy_backward(x.data_ptr(), y.data_ptr(), g, nc, d_loss.data_ptr(),
                                                      d_x.data_ptr(), _lib.current_stream(dev)),
                       "rwk_cross_entropy_backward")
        return d_x, None


def cross_entropy(scores: torch.Tensor, labels: torch.Tensor) -> torch.Tensor:
    """Mean cross-entropy; labels must lie in ``[0, C)`` (not checked on the device)."""
    return _CrossEntropy.apply(scores, labels)
