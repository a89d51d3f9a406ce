"""Fused classifier head: ``log_softmax(fc2(dropout(relu(fc1(bn(k))))))`` (``rwgnn2/model.py:68-72``).

The modules stay ordinary ``nn.BatchNorm1d`` / ``nn.Linear`` / ``nn.Dropout`` objects (same ``state_dict``);
this function only replaces the ~35 small launches torch needs for their forward and backward by two HIP
launches each way.  Batch statistics, running-statistics update and gradients follow torch's definitions; the
dropout mask comes from the model's counter-based generator instead of torch's.
"""
from __future__ import annotations

import torch

from . import _lib

__all__ = ["head_supported", "fused_head", "fused_head_loss"]


def head_supported(n_graphs: int, n_features: int, penultimate_dim: int, n_classes: int) -> bool:
    return bool(_lib.load().rwk_head_supported(n_graphs, n_features, penultimate_dim, n_classes))


def check_labels(labels, n_classes: int):
    """Class indices must lie in [0, n_classes), as ``F.cross_entropy`` requires (it raises otherwise; the kernels
    would clamp silently).  The range of a label tensor is read from the device once and kept ON the tensor object
    (with the version it was read at), so it lives exactly as long as the tensor and is read again after an in-place
    change; batches are reused every epoch (``rwgnn2/main.py:70-72``).  ``DeviceDataset.batch`` attaches the range of
    the labels it gathered, which it knows on the host: streamed batches cost no device read."""
    rng = getattr(labels, "_rwk_label_range", None)
    if rng is None or rng[0] != labels._version:
        if torch.cuda.is_current_stream_capturing():
            return                                        # no host read under graph capture; warm-up steps came first
        if labels.numel():
            lo, hi = torch.stack(torch.aminmax(labels)).cpu().tolist()
        else:
            lo, hi = 0, 0
        rng = (labels._version, int(lo), int(hi))
        labels._rwk_label_range = rng
    _, lo, hi = rng
    if lo < 0 or hi >= n_classes:
        raise IndexError(f"Target {lo if lo < 0 else hi} is out of bounds for {n_classes} classes")


def _launch_forward(lib, kc, g, c, p, nc, bn_wc, bn_bc, running_mean, running_var, num_batches_tracked, momentum, eps,
                    training, w1c, b1c, w2c, b2c, drop, rng_state, mean, invstd, h1d, out, labels, loss_partials,
                    handoff, stream):
    """One head forward.  ``handoff`` (from the random-walk layer's K reduction, see ``rw_kernel_layer``) carries the
    batch-statistics partials of ``kc`` and the drawn generator state: the statistics launch is then skipped."""
    dev = kc.device
    if training and handoff is not None and handoff.get("stats") is not None:
        rng_used = handoff.get("rng_used") if drop else None
        if drop and rng_used is None:
            raise RuntimeError("the statistics handoff does not carry the head's generator state")
        _lib.check(lib.rwk_head_apply_loss(
            kc.data_ptr(), g, c, p, nc, bn_wc.data_ptr(), bn_bc.data_ptr(), running_mean.data_ptr(),
            running_var.data_ptr(), _lib.ptr(num_batches_tracked), float(momentum), float(eps),
            handoff["stats"].data_ptr(), int(handoff["n_stats"]), w1c.data_ptr(), b1c.data_ptr(), w2c.data_ptr(),
            b2c.data_ptr(), drop, _lib.ptr(rng_used), _lib.ptr(mean), _lib.ptr(invstd), _lib.ptr(h1d), out.data_ptr(),
            _lib.ptr(labels), _lib.ptr(loss_partials), stream), "rwk_head_apply_loss")
        return
    stats = torch.empty(lib.rwk_head_stats_size(g, c), dtype=torch.float32, device=dev) if training else None
    rng_used = torch.empty(2, dtype=torch.int64, device=dev) if drop else None
    _lib.check(lib.rwk_head_forward_loss(
        kc.data_ptr(), g, c, p, nc, bn_wc.data_ptr(), bn_bc.data_ptr(), running_mean.data_ptr(),
        running_var.data_ptr(), _lib.ptr(num_batches_tracked), float(momentum), float(eps), int(training),
        _lib.ptr(stats), w1c.data_ptr(), b1c.data_ptr(), w2c.data_ptr(), b2c.data_ptr(), drop,
        _lib.ptr(rng_state) if drop else None, _lib.ptr(rng_used), _lib.ptr(mean), _lib.ptr(invstd),
        _lib.ptr(h1d), out.data_ptr(), _lib.ptr(labels), _lib.ptr(loss_partials), stream), "rwk_head_forward_loss")


def _split_grads(grads, c, p, nc):
    """(d_bn_w, d_bn_b, d_w1, d_b1, d_w2, d_b2) views of the kernels' gradient vector
    [dW2 (nc*p) | db2 (nc) | dW1 (p*c) | db1 (p) | dgamma (c) | dbeta (c)]."""
    o = 0
    d_w2 = grads[o:o + nc * p].view(nc, p); o += nc * p
    d_b2 = grads[o:o + nc]; o += nc
    d_w1 = grads[o:o + p * c].view(p, c); o += p * c
    d_b1 = grads[o:o + p]; o += p
    d_bn_w = grads[o:o + c]; o += c
    d_bn_b = grads[o:o + c]
    return d_bn_w, d_bn_b, d_w1, d_b1, d_w2, d_b2


class _FusedHead(torch.autograd.Function):
    @staticmethod
    def forward(ctx, k, bn_w, bn_b, w1, b1, w2, b2, running_mean, running_var, num_batches_tracked, momentum, eps,
                training, drop_p, rng_state, handoff=None):
        lib = _lib.load()
        if not k.is_cuda:
            raise RuntimeError("gnn_amd runs on HIP devices only (no CPU fallback)")
        g, c = k.shape
        p, nc = w1.shape[0], w2.shape[0]
        dev = k.device
        training = bool(training)
        if training and g < 2:
            raise ValueError("Expected more than 1 value per channel when training, got input size %s"
                             % (tuple(k.shape),))
        drop = float(drop_p) if (training and drop_p and rng_state is not None) else 0.0
        needs_grad = any(ctx.needs_input_grad[:7])
        with _lib.device_guard(dev):
            stream = _lib.current_stream(dev)
            kc = k.detach().contiguous()
            tensors = [t.detach().contiguous() for t in (bn_w, bn_b, w1, b1, w2, b2)]
            bn_wc, bn_bc, w1c, b1c, w2c, b2c = tensors
            out = torch.empty((g, nc), dtype=torch.float32, device=dev)
            keep = training or needs_grad
            mean = torch.empty(c, dtype=torch.float32, device=dev) if keep else None
            invstd = torch.empty(c, dtype=torch.float32, device=dev) if keep else None
            h1d = torch.empty((g, p), dtype=torch.float32, device=dev) if needs_grad else None
            _launch_forward(lib, kc, g, c, p, nc, bn_wc, bn_bc, running_mean, running_var, num_batches_tracked,
                            momentum, eps, training, w1c, b1c, w2c, b2c, drop, rng_state, mean, invstd, h1d, out,
                            None, None, handoff, stream)
        ctx.cfg = (training, drop)
        ctx.save_for_backward(kc, bn_wc, bn_bc, w1c, w2c, mean, invstd, h1d, out)
        return out

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, d_out):
        lib = _lib.load()
        kc, bn_wc, bn_bc, w1c, w2c, mean, invstd, h1d, out = ctx.saved_tensors
        training, drop = ctx.cfg
        g, c = kc.shape
        p, nc = w1c.shape[0], w2c.shape[0]
        dev = kc.device
        with _lib.device_guard(dev):
            stream = _lib.current_stream(dev)
            d_out = d_out.contiguous().float()
            nv = lib.rwk_head_grad_size(c, p, nc)
            blocks = lib.rwk_head_n_blocks(g)
            dy = torch.empty((g, c), dtype=torch.float32, device=dev)
            partials = torch.empty((blocks, nv), dtype=torch.float32, device=dev)
            grads = torch.empty(nv, dtype=torch.float32, device=dev)
            d_k = torch.empty((g, c), dtype=torch.float32, device=dev) if ctx.needs_input_grad[0] else None
            _lib.check(lib.rwk_head_backward(
                kc.data_ptr(), g, c, p, nc, bn_wc.data_ptr(), bn_bc.data_ptr(), mean.data_ptr(), invstd.data_ptr(),
                int(training), w1c.data_ptr(), w2c.data_ptr(), h1d.data_ptr(), out.data_ptr(), d_out.data_ptr(),
                drop, dy.data_ptr(), partials.data_ptr(), grads.data_ptr(), _lib.ptr(d_k), stream),
                "rwk_head_backward")
        o = 0
        d_w2 = grads[o:o + nc * p].view(nc, p); o += nc * p
        d_b2 = grads[o:o + nc]; o += nc
        d_w1 = grads[o:o + p * c].view(p, c); o += p * c
        d_b1 = grads[o:o + p]; o += p
        d_bn_w = grads[o:o + c]; o += c
        d_bn_b = grads[o:o + c]
        return (d_k, d_bn_w, d_bn_b, d_w1, d_b1, d_w2, d_b2) + (None,) * 9


class _FusedHeadLoss(torch.autograd.Function):
    """The head and ``F.cross_entropy(out, labels)`` (``rwgnn2/main.py:84-85``) as one autograd node: returns
    ``(out, loss)``; only ``loss`` carries a gradient.  The forward launch leaves per-workgroup loss sums, the
    backward launch forms the loss gradient in place and finishes the loss value, so the two cross-entropy
    launches disappear.  When no backward pass will follow, the loss is computed at once by its own kernel."""

    @staticmethod
    def forward(ctx, k, labels, bn_w, bn_b, w1, b1, w2, b2, running_mean, running_var, num_batches_tracked, momentum,
                eps, training, drop_p, rng_state, grad_mode, handoff=None):
        lib = _lib.load()
        if not k.is_cuda:
            raise RuntimeError("gnn_amd runs on HIP devices only (no CPU fallback)")
        g, c = k.shape
        p, nc = w1.shape[0], w2.shape[0]
        dev = k.device
        training = bool(training)
        if training and g < 2:
            raise ValueError("Expected more than 1 value per channel when training, got input size %s"
                             % (tuple(k.shape),))
        if labels.dtype != torch.int64 or labels.shape != (g,) or labels.device != dev:
            raise ValueError("labels must be an int64 tensor of one class index per graph on the device of k")
        drop = float(drop_p) if (training and drop_p and rng_state is not None) else 0.0
        check_labels(labels, nc)
        # `out` is marked non-differentiable below; autograd would still hand backward() a zero tensor of its shape
        # (a fill launch per step) unless told not to materialise undefined gradients
        ctx.set_materialize_grads(False)
        # grad_mode: the caller's torch.is_grad_enabled() (inside forward() it is always off, and
        # needs_input_grad reports requires_grad of the inputs whatever the mode)
        needs_grad = bool(grad_mode) and any(ctx.needs_input_grad[:1] + ctx.needs_input_grad[2:8])
        # training step: the tile part of the backward pass runs in the forward launch (rwk_head_train_forward)
        train_fused = needs_grad and training and bool(lib.rwk_head_train_supported(g, c, p, nc))
        with _lib.device_guard(dev):
            stream = _lib.current_stream(dev)
            kc = k.detach().contiguous()
            yc = labels.contiguous()
            bn_wc, bn_bc, w1c, b1c, w2c, b2c = [t.detach().contiguous() for t in (bn_w, bn_b, w1, b1, w2, b2)]
            out = torch.empty((g, nc), dtype=torch.float32, device=dev)
            loss = torch.empty((), dtype=torch.float32, device=dev)
            keep = training or needs_grad
            mean = torch.empty(c, dtype=torch.float32, device=dev) if keep else None
            invstd = torch.empty(c, dtype=torch.float32, device=dev) if keep else None
            loss_partials = torch.empty(lib.rwk_head_n_blocks(g), dtype=torch.float32, device=dev) if needs_grad \
                else None
            if train_fused:
                blocks = lib.rwk_head_n_blocks(g)
                dy = torch.empty((g, c), dtype=torch.float32, device=dev)
                partials = torch.empty((blocks, lib.rwk_head_grad_size(c, p, nc)), dtype=torch.float32, device=dev)
                ready = handoff is not None and handoff.get("stats") is not None
                if ready:
                    stats, n_stats, rng_used = handoff["stats"], int(handoff["n_stats"]), handoff.get("rng_used")
                    if drop and rng_used is None:
                        raise RuntimeError("the statistics handoff does not carry the head's generator state")
                else:
                    stats = torch.empty(lib.rwk_head_stats_size(g, c), dtype=torch.float32, device=dev)
                    n_stats = 0
                    rng_used = torch.empty(2, dtype=torch.int64, device=dev) if drop else None
                _lib.check(lib.rwk_head_train_forward(
                    kc.data_ptr(), g, c, p, nc, bn_wc.data_ptr(), bn_bc.data_ptr(), running_mean.data_ptr(),
                    running_var.data_ptr(), _lib.ptr(num_batches_tracked), float(momentum), float(eps),
                    stats.data_ptr(), n_stats, w1c.data_ptr(), b1c.data_ptr(), w2c.data_ptr(), b2c.data_ptr(), drop,
                    _lib.ptr(rng_state) if (drop and not ready) else None, _lib.ptr(rng_used) if drop else None,
                    mean.data_ptr(), invstd.data_ptr(), out.data_ptr(), yc.data_ptr(), loss_partials.data_ptr(),
                    dy.data_ptr(), partials.data_ptr(), loss.data_ptr(), stream), "rwk_head_train_forward")
                ctx.cfg = (training, drop, True)
                ctx.save_for_backward(kc, bn_wc, mean, invstd, dy, partials, loss_partials, loss)
                ctx.sizes = (p, nc)
                ctx.handoff = handoff
                ctx.mark_non_differentiable(out)
                return out, loss
            h1d = torch.empty((g, p), dtype=torch.float32, device=dev) if needs_grad else None
            _launch_forward(lib, kc, g, c, p, nc, bn_wc, bn_bc, running_mean, running_var, num_batches_tracked,
                            momentum, eps, training, w1c, b1c, w2c, b2c, drop, rng_state, mean, invstd, h1d, out,
                            yc if needs_grad else None, loss_partials, handoff, stream)
            if not needs_grad:
                _lib.check(lib.rwk_cross_entropy_forward(out.data_ptr(), yc.data_ptr(), g, nc, loss.data_ptr(),
                                                         stream), "rwk_cross_entropy_forward")
        ctx.cfg = (training, drop, False)
        ctx.save_for_backward(kc, yc, bn_wc, bn_bc, w1c, w2c, mean, invstd, h1d, out, loss_partials, loss)
        ctx.mark_non_differentiable(out)
        return out, loss

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, _d_out, d_loss):
        lib = _lib.load()
        training, drop, train_fused = ctx.cfg
        if d_loss is None:                                # nothing downstream used the loss
            return (None,) * 18
        if train_fused:
            kc, bn_wc, mean, invstd, dy, partials, loss_partials, loss = ctx.saved_tensors
            g, c = kc.shape
            p, nc = ctx.sizes
            dev = kc.device
            with _lib.device_guard(dev):
                stream = _lib.current_stream(dev)
                d_loss = d_loss.contiguous().float()
                grads = torch.empty(lib.rwk_head_grad_size(c, p, nc), dtype=torch.float32, device=dev)
                handoff = ctx.handoff
                if (handoff is not None and handoff.get("fold_ok") and handoff.get("tail") is not None
                        and ctx.needs_input_grad[0] and loss_partials is not None
                        and handoff["tail"].fold_allowed()):
                    # No launch here: the random-walk layer's backward (which runs next) forms d_k inside its fused
                    # kernel and sums the parameter gradients in the step's last launch (function.py, optim.StepTail).
                    # `grads` is filled by that launch, so autograd gets NO gradient for the six head parameters from
                    # this node (it would copy, accumulate or hand to hooks a buffer nobody has written yet): the
                    # step tail sets their .grad to views of `grads` once its launch is queued.  d_k is never read by
                    # the layer's backward in this shape; dy stands in for its shape.
                    handoff["head_fold"] = {"k": kc, "dy": dy, "mean": mean, "invstd": invstd, "bn_w": bn_wc,
                                            "partials": partials, "d_loss": d_loss, "loss_partials": loss_partials,
                                            # detached: the loss with its grad_fn here would close a reference
                                            # cycle (node -> hand-off -> loss -> node) and keep this step's autograd
                                            # graph, AccumulateGrad nodes included, alive into the next step
                                            "loss": loss.detach(), "grads": grads,
                                            "o_gamma": nc * p + nc + p * c + p,
                                            "views": _split_grads(grads, c, p, nc)}
                    handoff["head_grads"] = (grads, c, p, nc)
                    return (dy, None) + (None,) * 16
                d_k = torch.empty((g, c), dtype=torch.float32, device=dev) if ctx.needs_input_grad[0] else None
                _lib.check(lib.rwk_head_train_backward(
                    kc.data_ptr(), g, c, p, nc, bn_wc.data_ptr(), mean.data_ptr(), invstd.data_ptr(), dy.data_ptr(),
                    partials.data_ptr(), d_loss.data_ptr(), loss_partials.data_ptr(), loss.data_ptr(),
                    grads.data_ptr(), _lib.ptr(d_k), stream), "rwk_head_train_backward")
            if ctx.handoff is not None and ctx.handoff.get("tail") is not None:
                # the random-walk layer's backward (which runs next) updates the head's parameters from this buffer
                ctx.handoff["head_grads"] = (grads, c, p, nc)
            return (d_k, None) + _split_grads(grads, c, p, nc) + (None,) * 10
        kc, yc, bn_wc, bn_bc, w1c, w2c, mean, invstd, h1d, out, loss_partials, loss = ctx.saved_tensors
        g, c = kc.shape
        p, nc = w1c.shape[0], w2c.shape[0]
        dev = kc.device
        with _lib.device_guard(dev):
            stream = _lib.current_stream(dev)
            d_loss = d_loss.contiguous().float()
            nv = lib.rwk_head_grad_size(c, p, nc)
            blocks = lib.rwk_head_n_blocks(g)
            dy = torch.empty((g, c), dtype=torch.float32, device=dev)
            partials = torch.empty((blocks, nv), dtype=torch.float32, device=dev)
            grads = torch.empty(nv, dtype=torch.float32, device=dev)
            d_k = torch.empty((g, c), dtype=torch.float32, device=dev) if ctx.needs_input_grad[0] else None
            _lib.check(lib.rwk_head_backward_loss(
                kc.data_ptr(), g, c, p, nc, bn_wc.data_ptr(), bn_bc.data_ptr(), mean.data_ptr(), invstd.data_ptr(),
                int(training), w1c.data_ptr(), w2c.data_ptr(), h1d.data_ptr(), out.data_ptr(), None, drop,
                dy.data_ptr(), partials.data_ptr(), grads.data_ptr(), _lib.ptr(d_k), yc.data_ptr(),
                d_loss.data_ptr(), loss_partials.data_ptr(), loss.data_ptr(), stream), "rwk_head_backward_loss")
        o = 0
        d_w2 = grads[o:o + nc * p].view(nc, p); o += nc * p
        d_b2 = grads[o:o + nc]; o += nc
        d_w1 = grads[o:o + p * c].view(p, c); o += p * c
        d_b1 = grads[o:o + p]; o += p
        d_bn_w = grads[o:o + c]; o += c
        d_bn_b = grads[o:o + c]
        return (d_k, None, d_bn_w, d_bn_b, d_w1, d_b1, d_w2, d_b2) + (None,) * 10


def fused_head_loss(k, labels, bn, fc1, fc2, drop_p: float, training: bool, rng_state=None, handoff=None):
    """``out = fused_head(...)`` and ``loss = F.cross_entropy(out, labels)`` in the head's own launches; returns
    ``(out, loss)`` with ``out`` detached.  With gradients enabled the value of ``loss`` is complete once
    ``loss.backward()`` has run (the backward launch finishes the sum): read it after the step."""
    return _FusedHeadLoss.apply(k, labels, bn.weight, bn.bias, fc1.weight, fc1.bias, fc2.weight, fc2.bias,
                                bn.running_mean, bn.running_var, bn.num_batches_tracked, bn.momentum, bn.eps,
                                training, drop_p, rng_state, torch.is_grad_enabled(), handoff)


def fused_head(k, bn, fc1, fc2, drop_p: float, training: bool, rng_state=None, handoff=None):
    """``log_softmax(fc2(dropout(relu(fc1(bn(k))))))`` with ``bn`` an ``nn.BatchNorm1d`` (affine, tracking running
    statistics, float momentum) and ``fc1``/``fc2`` ``nn.Linear`` with bias."""
    return _FusedHead.apply(k, bn.weight, bn.bias, fc1.weight, fc1.bias, fc2.weight, fc2.bias, bn.running_mean,
                            bn.running_var, bn.num_batches_tracked, bn.momentum, bn.eps, training, drop_p, rng_state,
                            handoff)
