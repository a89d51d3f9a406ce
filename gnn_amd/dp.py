"""Graph-sharded data parallelism: one process per GPU, parameters replicated, ONE gradient all-reduce per step.

Graphs of a batch are independent (block-diagonal adjacency; row g of the kernel features depends on
graph g only), so a batch shards by contiguous ranges of graphs with no collective in the data path.
The only exchange is the gradient sum over a single flat fp32 bucket (about 1.7k floats for the default
hyper-parameters, 49k for 64 hidden graphs of 10 nodes at hidden_dim 64): a latency-bound message, so it
is issued as one RCCL all-reduce rather than per-parameter or bucketed-and-overlapped calls.

The reference has no multi-device path; batch-norm statistics stay per rank.
"""
from __future__ import annotations

import os
from typing import List, Sequence, Tuple

import numpy as np
import torch
import torch.distributed as dist

__all__ = ["shard_ranges", "FlatGradBucket", "broadcast_parameters"]


def shard_ranges(weights: Sequence[int], world_size: int) -> List[Tuple[int, int]]:
    """Split graphs 0..G-1 into ``world_size`` contiguous ranges of nearly equal total weight
    (weight = nodes or edges per graph).  Every range is non-empty when G >= world_size."""
    w = np.asarray(weights, dtype=np.float64)
    g = w.size
    if world_size < 1:
        raise ValueError("world_size must be positive")
    if g < world_size:
        raise ValueError(f"cannot shard {g} graphs over {world_size} ranks")
    csum = np.concatenate([[0.0], np.cumsum(w)])
    bounds = [0]
    for r in range(1, world_size):
        target = csum[-1] * r / world_size
        cut = int(np.searchsorted(csum, target, side="left"))
        cut = max(cut, bounds[-1] + 1)              # non-empty
        cut = min(cut, g - (world_size - r))        # leave one graph for each later rank
        bounds.append(cut)
    bounds.append(g)
    return [(bounds[i], bounds[i + 1]) for i in range(world_size)]


class FlatGradBucket:
    """All parameter gradients of a module as views into one contiguous fp32 buffer.

    ``p.grad`` of every parameter aliases a slice of ``self.flat``; autograd accumulates into those views
    in place, ``zero()`` clears the buffer with one fill and ``all_reduce()`` issues one collective.
    """

    def __init__(self, module, bind_grads: bool = True):
        """``module``: an ``nn.Module`` or an iterable of parameters.  ``bind_grads=False`` leaves the parameters'
        ``.grad`` alone (a caller that gathers them with :meth:`gather` or inside the optimiser's launch)."""
        params = module.parameters() if isinstance(module, torch.nn.Module) else module
        self.params = [p for p in params if p.requires_grad]
        if not self.params:
            raise ValueError("module has no trainable parameters")
        dev, dtype = self.params[0].device, self.params[0].dtype
        total = sum(p.numel() for p in self.params)
        self.flat = torch.zeros(total, dtype=dtype, device=dev)
        self._avg_ok = None                # whether the backend accepted ReduceOp.AVG (unknown until first use)
        off = 0
        for p in self.params if bind_grads else ():
            n = p.numel()
            p.grad = self.flat[off:off + n].view_as(p)
            off += n

    def zero(self):
        self.flat.zero_()

    def release(self):
        """Detach ``p.grad`` from the bucket (host-side only): the next backward then hands each gradient over
        without an accumulation launch per parameter; :meth:`gather` packs them into ``flat`` with one launch."""
        for p in self.params:
            p.grad = None

    def gather(self):
        """``flat`` = concatenation of the parameters' current gradients (zeros for a parameter without one)."""
        grads = [None if p.grad is None else p.grad.detach().reshape(-1) for p in self.params]
        if self.flat.is_cuda:
            import ctypes
            from . import _lib
            lib = _lib.load()
            grads = [None if g is None else g.contiguous().float() for g in grads]
            n = len(grads)
            srcs = (ctypes.c_void_p * n)(*[None if g is None else g.data_ptr() for g in grads])
            sizes = (ctypes.c_int64 * n)(*[p.numel() for p in self.params])
            with _lib.device_guard(self.flat.device):
                _lib.check(lib.rwk_gather_chunks(srcs, sizes, n, self.flat.data_ptr(),
                                                 _lib.current_stream(self.flat.device)), "rwk_gather_chunks")
            self._keep = grads                     # the launch reads them asynchronously
        else:                                       # host tensors (process-group tests on the gloo backend)
            off = 0
            for p, g in zip(self.params, grads):
                n = p.numel()
                if g is None:
                    self.flat[off:off + n].zero_()
                else:
                    self.flat[off:off + n].copy_(g)
                off += n

    def all_reduce(self, group=None, average: bool = True):
        """Sum (or mean) of the bucket over the ranks of ``group``; a no-op for a single process."""
        if not dist.is_available() or not dist.is_initialized():
            return
        world = dist.get_world_size(group)
        if world == 1 and os.environ.get("RWK_REHEARSE_RCCL", "0") != "1":
            return       # (the rehearsal switch issues the collective on a one-rank group: bench.py)
        if average and self.flat.is_cuda and dist.get_backend(group) == "nccl" and self._avg_ok is not False:
            # RCCL averages inside the collective: no separate division launch
            try:
                dist.all_reduce(self.flat, op=dist.ReduceOp.AVG, group=group)
                self._avg_ok = True
                return
            except (RuntimeError, ValueError):
                if self._avg_ok:                      # it worked before: a real failure, not a missing feature
                    raise
                self._avg_ok = False
        dist.all_reduce(self.flat, op=dist.ReduceOp.SUM, group=group)
        if average:
            self.flat.div_(world)


def broadcast_parameters(module: torch.nn.Module, src: int = 0, group=None):
    """Make every rank start from rank ``src``'s parameters and buffers."""
    if not dist.is_available() or not dist.is_initialized() or dist.get_world_size(group) == 1:
        return
    for t in list(module.parameters()) + list(module.buffers()):
        dist.broadcast(t.data, src=src, group=group)
