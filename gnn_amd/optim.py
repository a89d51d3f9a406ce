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

__all__ = ["FlatAdam", "StepTail", "Adam"]


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


class Adam(torch.optim.Optimizer):
    """``torch.optim.Adam`` for the reference's training loop (``rwgnn2/main.py:79``: ``optim.Adam(model.parameters(),
    lr=args.lr)``) with the whole update in ONE launch: ``optimizer = gnn_amd.Adam(model.parameters(), lr=args.lr)``.

    A ``torch.optim.Optimizer`` like any other - ``zero_grad()``, ``step()``, ``param_groups`` (a learning-rate
    scheduler such as ``StepLR``, ``main.py:80``, changes ``param_groups[0]["lr"]`` and the next step uses it),
    ``state_dict()`` / ``load_state_dict()`` in torch's layout (``main.py:133,146``).  Same arithmetic as
    ``torch.optim.Adam`` without amsgrad / maximize.  torch's own implementation spends 0.13-0.24 ms of host time per
    step on the ten small tensors of ``RW_NN`` (some forty launches); this one gathers the ``.grad`` tensors and
    updates the flat parameter buffer in one (:class:`FlatAdam`).  The parameters become views of one flat buffer
    (same values, same objects).  One parameter group, all parameters on one HIP device."""

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0):
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))
        if len(self.param_groups) != 1:
            raise ValueError("gnn_amd.Adam takes one parameter group")
        group = self.param_groups[0]
        bucket = FlatGradBucket([p for p in group["params"]], bind_grads=False)
        if len(bucket.params) != len(group["params"]):
            raise ValueError("every parameter must require a gradient")
        self._flat = FlatAdam(None, bucket, lr=group["lr"], betas=group["betas"], eps=group["eps"],
                              weight_decay=group["weight_decay"])
        self._lr_on_device = float(group["lr"])
        off = 0
        for p in bucket.params:                   # torch's layout of the state, as views of the flat buffers
            n = p.numel()
            self.state[p] = {"step": torch.zeros((), dtype=torch.float32),
                             "exp_avg": self._flat.exp_avg[off:off + n].view_as(p),
                             "exp_avg_sq": self._flat.exp_avg_sq[off:off + n].view_as(p)}
            off += n

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        group = self.param_groups[0]
        lr = float(group["lr"])
        if lr != self._lr_on_device:              # a scheduler stepped: one fill, only then
            self._flat.set_lr(lr)
            self._lr_on_device = lr
        self._flat.betas = (float(group["betas"][0]), float(group["betas"][1]))
        self._flat.eps = float(group["eps"])
        self._flat.weight_decay = float(group["weight_decay"])
        self._flat.step(gather=True)
        return loss

    def state_dict(self):
        # the step count lives on the device (it is read here only: checkpoints, main.py:133)
        t = float(self._flat.step_counters[0].item())
        for st in self.state.values():
            st["step"] = torch.tensor(t, dtype=torch.float32)
        return super().state_dict()

    def load_state_dict(self, state_dict):
        super().load_state_dict(state_dict)       # leaves freshly copied tensors in self.state: move them back
        off = 0
        steps = []
        for p in self._flat.bucket.params:
            n = p.numel()
            st = self.state.get(p, {})
            for name, flat in (("exp_avg", self._flat.exp_avg), ("exp_avg_sq", self._flat.exp_avg_sq)):
                view = flat[off:off + n].view_as(p)
                if name in st:
                    view.copy_(st[name])
                else:
                    view.zero_()
                st[name] = view
            steps.append(float(st.get("step", 0.0)))
            st.setdefault("step", torch.zeros((), dtype=torch.float32))
            self.state[p] = st
            off += n
        if len(set(steps)) > 1:
            raise ValueError("gnn_amd.Adam keeps one step count for all parameters")
        self._flat.step_counters.fill_(int(steps[0]) if steps else 0)
        self._lr_on_device = None                 # param_groups was replaced: push the rate at the next step


class StepTail:
    """The optimiser's update inside the backward pass's last launch (single-process training step).

    ``RW_NN``'s backward ends with one small launch (hidden-graph backward + the input projection's gradient sum) and
    the step with another (Adam over the flat buffer).  Adam is element-wise, so the threads of the former can apply
    it to the gradient elements they finish and a few more workgroups to the head's parameters, whose gradients are
    in memory by then (``rwk_hidden_backward_reduce_adam``): one launch less per step, the same arithmetic.

    ``Trainer`` hands an instance to the model (``model._step_tail``); the model passes it down with the hand-off
    dictionary, the head's backward leaves its gradient buffer there, and the layer's backward calls :meth:`launch`
    in place of ``rwk_hidden_backward_reduce``.  ``done`` then tells the trainer to skip ``optimizer.step()``.
    Requires the flat buffer to start with ``adj_hidden, features_hidden, fc.weight, fc.bias`` (``RW_NN``'s
    parameter order) followed by the head's six tensors.

    When the head's finishing launch is folded in as well (``head_fold``), the gradients of the head's six parameters
    do not travel through autograd at all: this launch computes them, and :meth:`launch` then points the parameters'
    ``.grad`` at them.  They are valid once that launch has run, i.e. for everything queued on the stream after the
    backward pass - not inside tensor hooks of the backward pass itself, which is why the fold is taken only when
    those parameters carry no hooks and no gradient to accumulate into (:meth:`fold_allowed`)."""

    def __init__(self, model, opt: FlatAdam, update: bool = True):
        self.opt = opt
        # update=False (data-parallel step): the launch only completes the flat gradient bucket - no gather launch,
        # no finishing launch of the head - and the all-reduce and the stand-alone Adam launch follow
        self.update = bool(update)
        self.done = False
        self.armed = False        # set by the trainer around its own forward + backward only
        self._shape = None
        self.head_folds = 0       # launches that also summed the head's parameter gradients (no finishing launch)
        params = opt.bucket.params
        want = [model.adj_hidden, model.features_hidden, model.fc.weight, model.fc.bias, model.bn.weight,
                model.bn.bias, model.fc1.weight, model.fc1.bias, model.fc2.weight, model.fc2.bias]
        if len(params) != len(want) or any(a is not b for a, b in zip(params, want)):
            raise ValueError("StepTail needs RW_NN's ten parameters, all trainable, in the module's order")
        sizes = [p.numel() for p in params]
        self.off_adj, self.off_feat, self.off_red = 0, sizes[0], sizes[0] + sizes[1]
        self.red_v = sizes[2] + sizes[3]
        self.rest_off = self.off_red + self.red_v
        self.n = sum(sizes)
        self.rest_sizes = sizes[4:]
        lib = _lib.load()
        n_hidden = model.adj_hidden.shape[0]
        n_counters = max(int(lib.rwk_step_tail_n_counters(n_hidden, self.n, self.rest_off)),
                         int(lib.rwk_step_tail_n_counters_head(n_hidden, self.n, self.rest_off)))
        dev = opt.flat_param.device
        # hand-over of Adam's bias corrections from one step's launch to the next (rwk.h: corr_slot);
        # RWK_NO_CORR_SLOT=1 switches it off (every workgroup then computes its own, as the stand-alone kernels do)
        import os
        self.corr_slot = None if os.environ.get("RWK_NO_CORR_SLOT") == "1" else torch.zeros(4, dtype=torch.int64,
                                                                                              device=dev)
        # one counter per workgroup of either launch shape, all at the optimiser's step count: the two shapes may
        # then alternate (the counters one shape does not touch are brought along by sync_counters)
        if not self.update:
            self.counters = None          # the stand-alone Adam launch keeps its own
            return
        self.counters = torch.zeros(max(n_counters, opt.step_counters.numel()), dtype=torch.int64, device=dev)
        self.counters.copy_(opt.step_counters[:1].expand_as(self.counters))
        opt.step_counters = self.counters

    def fold_allowed(self) -> bool:
        """The head's six parameters have nothing that would consume a gradient from autograd during the backward
        pass: no ``.grad`` to accumulate into, no tensor hooks, no post-accumulate hooks."""
        for p in self.opt.bucket.params[4:]:
            if p.grad is not None or getattr(p, "_backward_hooks", None) \
                    or getattr(p, "_post_accumulate_grad_hooks", None):
                return False
        return True

    def sync_counters(self):
        """After an update by the stand-alone Adam launch (which advances only its own workgroups' counters)."""
        if self.counters is not None:
            self.counters.copy_(self.counters[:1].expand_as(self.counters))

    def head_sources(self, head_grads: torch.Tensor, c: int, p: int, nc: int):
        """Device addresses of the gradients of bn.weight, bn.bias, fc1.weight, fc1.bias, fc2.weight, fc2.bias inside
        the head's gradient buffer ([w2 | b2 | w1 | b1 | gamma | beta])."""
        base, f = head_grads.data_ptr(), 4
        o_w2, o_b2 = 0, nc * p
        o_w1 = o_b2 + nc
        o_b1 = o_w1 + p * c
        o_ga = o_b1 + p
        o_be = o_ga + c
        if [c, c, p * c, p, nc * p, nc] != self.rest_sizes:
            raise ValueError("head gradient buffer does not match the optimiser's parameter sizes")
        return [base + f * o for o in (o_ga, o_be, o_w1, o_b1, o_w2, o_b2)]

    def head_columns(self, c: int, p: int, nc: int):
        """Columns of the head's partial rows (layout [w2 | b2 | w1 | b1 | gamma | beta]) at which the gradients of
        bn.weight, bn.bias, fc1.weight, fc1.bias, fc2.weight, fc2.bias start."""
        o_w2, o_b2 = 0, nc * p
        o_w1 = o_b2 + nc
        o_b1 = o_w1 + p * c
        o_ga = o_b1 + p
        o_be = o_ga + c
        if [c, c, p * c, p, nc * p, nc] != self.rest_sizes:
            raise ValueError("head gradient buffer does not match the optimiser's parameter sizes")
        return [o_ga, o_be, o_w1, o_b1, o_w2, o_b2]

    def before_launch(self, shape: str):
        """Bookkeeping in front of a launch of shape ``"head"`` (the head's parameter-gradient sums ride along) or
        ``"plain"``."""
        o = self.opt
        # the handed-over bias corrections are those of one (beta1, beta2) pair: other betas start a fresh slot
        if self.corr_slot is not None and getattr(self, "_slot_betas", o.betas) != o.betas:
            self.corr_slot.zero_()
        self._slot_betas = o.betas
        # the two launch shapes have different workgroup counts, each workgroup advancing its own counter: when the
        # shape changes, the counters the previous shape did not touch are brought along first
        if self.update and self._shape is not None and self._shape != shape:
            self.sync_counters()
        self._shape = shape

    def after_launch(self, head_fold: bool):
        """The launch was queued by someone else (``composite.TrainStep``)."""
        self.head_folds += int(head_fold)
        self.done = True

    def launch(self, lib, stream, head_srcs, adj_hidden, ah, z, dz, dz_rows, n_hidden, s, d, steps, d_adj, d_feat,
               partials, blocks, red_v, red_out, head_fold=None, n_graphs=0, head_columns=None):
        import ctypes
        if red_v != self.red_v:
            raise ValueError("projection gradient width does not match fc.weight and fc.bias")
        o = self.opt
        n = len(head_srcs)
        sizes = (ctypes.c_int64 * n)(*self.rest_sizes)
        upd = (lambda t: t.data_ptr()) if self.update else (lambda t: None)
        self.before_launch("head" if head_fold is not None else "plain")
        if head_fold is not None:
            # the head's finishing launch was left out: its parameter gradients are summed here from its partial rows
            hp_ = head_fold["partials"]
            cols = (ctypes.c_int64 * n)(*head_columns)
            _lib.check(lib.rwk_hidden_backward_reduce_adam_head(
                adj_hidden, ah, z, dz, dz_rows, n_hidden, s, d, steps, d_adj, d_feat, partials, blocks, red_v, red_out,
                upd(o.flat_param), upd(o.exp_avg), upd(o.exp_avg_sq), o.bucket.flat.data_ptr(),
                self.n, self.off_adj, self.off_feat, self.off_red, cols, sizes, n, self.rest_off, upd(o.lr),
                o.betas[0], o.betas[1], o.eps, o.weight_decay, upd(self.counters), _lib.ptr(self.corr_slot),
                hp_.data_ptr(), hp_.shape[0],
                hp_.shape[1], head_fold["d_loss"].data_ptr(), head_fold["grads"].data_ptr(),
                head_fold["loss_partials"].data_ptr(), head_fold["loss"].data_ptr(), int(n_graphs), stream),
                "rwk_hidden_backward_reduce_adam_head")
            # autograd handed these six parameters nothing (head.py): their gradients are what this launch writes
            for prm, view in zip(self.opt.bucket.params[4:], head_fold["views"]):
                prm.grad = view
            self.head_folds += 1
            self.done = True
            return
        srcs = (ctypes.c_void_p * n)(*head_srcs)
        _lib.check(lib.rwk_hidden_backward_reduce_adam(
            adj_hidden, ah, z, dz, dz_rows, n_hidden, s, d, steps, d_adj, d_feat, partials, blocks, red_v, red_out,
            upd(o.flat_param), upd(o.exp_avg), upd(o.exp_avg_sq), o.bucket.flat.data_ptr(), self.n,
            self.off_adj, self.off_feat, self.off_red, srcs, sizes, n, self.rest_off, upd(o.lr), o.betas[0],
            o.betas[1], o.eps, o.weight_decay, upd(self.counters), _lib.ptr(self.corr_slot), stream),
            "rwk_hidden_backward_reduce_adam")
        self.done = True
