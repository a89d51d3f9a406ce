"""The reference's training step (``rwgnn2/main.py:82-88``) with the launch overhead taken out.

``train()`` in the reference is: ``optimizer.zero_grad()``, ``model(adj, features, graph_indicator)``,
``F.cross_entropy``, ``backward()``, ``optimizer.step()``.  On an MI355X the kernels of one such step on a
1024-graph batch take about 0.2 ms, less than the host needs to launch them one by one, so ``Trainer`` offers
the same step (a) eagerly and (b) captured once per batch into a HIP graph (two under data parallelism:
forward+backward and optimiser, with the gradient all-reduce between them) that is replayed every epoch; batches are reused every epoch in the reference (``main.py:70-72,103-106``),
which is what makes capture worthwhile.  Gradients live in one flat bucket (one all-reduce under data
parallelism, one Adam launch).  The dropout generator's state lives on the device, so every replay draws
fresh masks.
"""
from __future__ import annotations

import os

import contextlib
import warnings

import torch
import torch.nn.functional as F

from .dp import FlatGradBucket
from .losses import cross_entropy
from .optim import FlatAdam, StepTail
from .structure import get_structure

__all__ = ["Trainer", "EpochTrainer", "StreamTrainer", "stale_autograd_graph_is_an_error"]

_STALE_MESSAGE = "The AccumulateGrad node's stream does not match"


@contextlib.contextmanager
def stale_autograd_graph_is_an_error():
    """Inside this block, a backward pass that meets ``AccumulateGrad`` nodes created on another stream raises
    ``RuntimeError`` instead of torch's warn-once message.  Such nodes survive only while something still refers to an
    earlier step's autograd graph (a kept ``loss``, for instance).  They are harmless to an eager step, but they fork
    a backward pass that is being captured over two streams, and ending such a capture has crashed the process
    (a segmentation fault inside ``capture_end``, not an exception).  ``Trainer.capture`` therefore runs its warm-up
    steps - on a side stream, like the capture itself - and the capture under this check: a graph that is still
    alive is reported by the first warm-up step, before any capture has begun."""
    prev = torch.is_warn_always_enabled()
    torch.set_warn_always(True)              # the message is a warn-once: make every occurrence count
    try:
        with warnings.catch_warnings():
            warnings.filterwarnings("error", message=_STALE_MESSAGE)
            yield
    except UserWarning as w:
        if _STALE_MESSAGE not in str(w):
            raise
        raise RuntimeError(
            "an autograd graph of an earlier step is still alive (is a loss tensor of an earlier step, or anything "
            "else with a grad_fn that reaches the parameters, still referenced?): its AccumulateGrad nodes belong to "
            "another stream and would fork the captured backward pass over two streams; drop the reference "
            "(or .detach() what is kept) and capture again") from w
    finally:
        torch.set_warn_always(prev)


class Trainer:
    """One model, one resident batch ``(adj, features, graph_indicator, y)``.

    ``world`` > 1: the flat gradient bucket is all-reduced (mean) over the default process group between the
    backward and the optimiser step.  ``fused_loss``: ``gnn_amd.losses.cross_entropy`` (same value and gradient
    as ``F.cross_entropy``, two launches instead of torch's eight; for ``RW_NN`` inside the head's own launches).  ``optimizer``: "flat" (``gnn_amd.optim.FlatAdam``, same arithmetic as
    ``torch.optim.Adam``, one launch) or "torch" (``torch.optim.Adam(capturable=True)``).
    """

    def __init__(self, model, batch_t, world: int = 1, optimizer: str = "flat", lr: float = 1e-2,
                 fused_loss: bool = True, capture_error_mode: str = None, sync_before_all_reduce: bool = False,
                 forward_fn=None, capture_collective: bool = False, merge_tail: bool = True):
        self.model = model
        self.adj, self.feats, self.gi, self.y = batch_t
        # forward_fn(): the model's scores for the resident batch, for models with another call signature than
        # RW_NN's (gnn_amd.rw_gnn.RW_GNN takes one ``data`` object); its lazily created layers must exist already
        self.forward_fn_given = forward_fn is not None
        self._train_step = None
        self.forward_fn = forward_fn if forward_fn is not None else (
            lambda: self.model(self.adj, self.feats, self.gi))
        self.bucket = FlatGradBucket(model)
        self.world = world
        if optimizer == "flat":       # main.py:79: Adam, lr 1e-2
            self.opt = FlatAdam(model, self.bucket, lr=lr)
        else:
            self.opt = torch.optim.Adam(model.parameters(), lr=lr, capturable=True, foreach=True)
        # F.cross_entropy(output, y) of main.py:85, one launch each way (gnn_amd.losses) or torch's own kernels
        self.loss_fn = cross_entropy if fused_loss else F.cross_entropy
        # RW_NN with the fused loss: the cross-entropy rides in the head's launches (model.forward_with_loss); the
        # loss value is then complete at the end of the step, which is when step_*() callers read it
        self.loss_in_head = fused_loss and forward_fn is None and hasattr(model, "forward_with_loss")
        # one process, flat optimiser: the gradient gather rides in the Adam launch
        self.fused_gather = world == 1 and optimizer == "flat"
        # ... and for RW_NN the Adam update itself rides in the backward pass's last launch (optim.StepTail), whenever
        # the step takes the fused path (narrow inputs, hidden_dim 4, at least two steps, fused head with the loss)
        # Data parallel: the same launch without the update - it completes the flat gradient bucket (no gather launch,
        # no finishing launch of the head); the all-reduce and the stand-alone Adam launch follow.
        self.tail = None
        if merge_tail and optimizer == "flat" and self.loss_in_head:
            try:
                self.tail = StepTail(model, self.opt, update=world == 1)
                model._step_tail = self.tail
            except (ValueError, AttributeError):
                self.tail = None
        # gloo's all-reduce of a device tensor was seen to hang when the device still has queued work (two ranks on
        # one GPU, with or without graphs); rehearsals on that backend wait for the device first.  RCCL runs do not.
        self.sync_before_all_reduce = sync_before_all_reduce
        # Under data parallelism the collective library's watchdog thread polls events while the step is being
        # captured; "thread_local" keeps its calls from invalidating the capture.
        self.capture_error_mode = capture_error_mode or ("thread_local" if world > 1 else "global")
        # world > 1: try to capture the gradient all-reduce inside the step's graph (one replay per step, as with a
        # single process); a capture the collective library refuses falls back to the two-graph shape
        self.capture_collective = bool(capture_collective)
        self.collective_captured = False
        self.capture_note = ""
        self.loss = None
        # d(loss)/d(loss) = 1, kept resident: backward() would otherwise launch a fill for it every step
        self._one = torch.ones((), dtype=torch.float32, device=self.y.device)
        self.g_fb = None
        self.g_opt = None
        self.opt_in_graph = False
        self.g_multi = None
        self.multi_k = 1
        self.losses = []
        self._single_loss = None

    def _one_call(self) -> bool:
        """The whole step as one library call when the model, the batch and the optimiser allow it
        (``composite.TrainStep``); False: nothing was launched."""
        if self.tail is None or self.forward_fn_given:
            return False
        model = self.model
        if not hasattr(model, "_head_fusable") or not self.adj.is_cuda:
            return False
        if self._train_step is None:
            from .composite import TrainStep
            self._train_step = TrainStep(self)
        st = get_structure(self.adj, self.gi)
        drop = float(model._modules["dropout"].p) if model.training else 0.0
        entry = self._train_step.supported(model, st, self.feats, self.y, drop)
        if entry is None:
            return False
        self.loss = self._train_step.run(entry, model, st, self.feats, self.y)
        if not self.tail.update:
            self.tail.done = False       # the flat gradient is in the bucket; all-reduce and update follow
        return True

    def fwd_bwd(self):
        # the previous step's autograd graph goes first: while it lives, so do its AccumulateGrad nodes, and the new
        # forward would reuse them, whatever stream they were created on
        self.loss = None
        if self._one_call():
            return
        self.bucket.release()            # optimizer.zero_grad(set_to_none=True)
        if self.tail is not None:
            self.tail.armed = True       # this backward pass ends with the optimiser's update
        try:
            if self.loss_in_head:
                _, self.loss = self.model.forward_with_loss(self.adj, self.feats, self.gi, self.y)
            else:
                out = self.forward_fn()
                self.loss = self.loss_fn(out, self.y)
            self.loss.backward(self._one)
        finally:
            if self.tail is not None:
                self.tail.armed = False
        if self.tail is not None and not self.tail.update and self.tail.done:
            self.tail.done = False       # the backward pass's last launch left the flat gradient in the bucket
        elif not self.fused_gather:
            self.bucket.gather()         # per-parameter gradients -> the flat bucket, one launch

    def step_eager(self):
        self.fwd_bwd()
        if self.world > 1:
            self._all_reduce()
        self._opt_step()

    def _all_reduce(self):
        if self.sync_before_all_reduce:
            torch.cuda.current_stream().synchronize()
        self.bucket.all_reduce()

    def _opt_step(self):
        if self.tail is not None and self.tail.done:      # the update rode in the backward pass's last launch
            self.tail.done = False
            return
        if self.fused_gather:
            self.opt.step(gather=True)
            if self.tail is not None:
                self.tail.sync_counters()
        else:
            self.opt.step()

    def capture(self, steps_per_graph: int = 1):
        """Capture the step as HIP graphs.  One process: one graph.  Data parallel: forward+backward and the Adam
        update as two graphs with the (tiny) gradient all-reduce an ordinary RCCL call on the same stream between
        them - or, with ``capture_collective``, the all-reduce inside a single graph when the collective library
        accepts being captured (checked by replaying it once against the eager collective; any refusal or mismatch
        falls back to the two-graph shape and is recorded in ``capture_note``).

        ``steps_per_graph`` > 1 (one process): a second graph holds that many consecutive steps, for
        :meth:`run_steps`.  Starting a replay costs about as much as four kernel boundaries inside one (7.5 us
        against 1.75 us on an MI355X), so a loop that replays several steps at a time - an epoch's batches in the
        reference's loop, ``main.py:103-106`` - pays it once per replay instead of once per step.  Everything a step
        carries over (parameters, moments, step count, dropout generator state, running statistics) lives on the
        device, so the result is the same as that many single-step replays."""
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        # Warm-up on a side stream, under the stale-graph check: an autograd graph of an earlier step that something
        # still refers to raises here, before any capture has begun (see stale_autograd_graph_is_an_error).
        with torch.cuda.stream(s), stale_autograd_graph_is_an_error():
            for _ in range(3):
                self.step_eager()
        torch.cuda.current_stream().wait_stream(s)
        torch.cuda.synchronize()
        # Drop the warm-up's autograd graph before capturing: its AccumulateGrad nodes are bound to the warm-up
        # stream, and reusing them would fork the captured backward over two streams (the forked replay was seen to
        # feed stale gradients to the random-walk layer, and ending such a capture to crash the process).
        self.loss = None
        if self.world > 1 and self.capture_collective and not self.sync_before_all_reduce:
            if self._capture_single_graph():
                # with the collective inside the graph nothing of a step happens on the host: several steps per replay,
                # as with a single process
                if steps_per_graph > 1:
                    self._capture_multi(steps_per_graph, collective=True)
                return
        self.collective_captured = False
        self.g_fb = torch.cuda.CUDAGraph()
        with stale_autograd_graph_is_an_error():
            with torch.cuda.graph(self.g_fb, capture_error_mode=self.capture_error_mode):
                self.fwd_bwd()
                if self.world == 1:          # nothing happens between backward and the update: one graph
                    self._opt_step()
        self._single_loss = self.loss = self.loss.detach()   # same storage, without the captured autograd graph
        if self.world > 1:
            # The flat Adam update is ONE launch: as a graph of its own it would pay a replay's start (7.5 us) for a
            # 2.6 us kernel, so it is launched directly after the all-reduce.  torch.optim.Adam's dozens of launches
            # are worth a second graph.
            self.opt_in_graph = not isinstance(self.opt, FlatAdam)
            if self.opt_in_graph:
                self.g_opt = torch.cuda.CUDAGraph()
                with torch.cuda.graph(self.g_opt, capture_error_mode=self.capture_error_mode):
                    self._opt_step()
        elif steps_per_graph > 1:
            self._capture_multi(steps_per_graph)

    def _capture_multi(self, steps_per_graph: int, collective: bool = False):
        """A second graph of ``steps_per_graph`` consecutive steps (``collective``: each with its all-reduce)."""
        g_multi, losses = torch.cuda.CUDAGraph(), []
        try:
            with torch.cuda.graph(g_multi, capture_error_mode=self.capture_error_mode):
                for _ in range(steps_per_graph):
                    self.fwd_bwd()
                    if collective:
                        self.bucket.all_reduce()
                    self._opt_step()
                    losses.append(self.loss.detach())
            self.g_multi, self.losses, self.multi_k = g_multi, losses, int(steps_per_graph)
        except Exception as exc:      # the single-step graph stays usable
            torch.cuda.synchronize()
            note = "multi-step capture refused (%s: %s)" % (type(exc).__name__, str(exc)[:200])
            self.capture_note = (self.capture_note + "; " if self.capture_note else "") + note
        if collective and not self._all_ranks(self.multi_k == int(steps_per_graph)):
            # some rank has no multi-step graph: every rank replays single steps, so that all of them issue the
            # collectives from the same kind of launch
            if self.multi_k > 1:
                self.capture_note += "; multi-step graph dropped (another rank refused the capture)"
            self.g_multi, self.losses, self.multi_k = None, [], 1
        self.loss = self._single_loss  # step_graph() leaves its loss here; run_steps() points it at the last one

    def _all_ranks(self, ok: bool) -> bool:
        """True when ``ok`` holds on every rank (one eager MIN all-reduce; every rank must call it at the same point,
        whatever its own outcome, so that a capture refused on some ranks only cannot leave the others waiting in a
        collective nobody else issues)."""
        import torch.distributed as dist
        if not (dist.is_available() and dist.is_initialized()):
            return bool(ok)
        flag = torch.tensor([1 if ok else 0], dtype=torch.int32, device=self.bucket.flat.device)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        return bool(int(flag.item()))

    def _capture_single_graph(self) -> bool:
        """forward+backward, all-reduce and update in one graph.  Returns False (with everything as it was before,
        the reason in ``capture_note``) when the capture raises or when a replay does not leave the averaged
        gradient in the bucket."""
        g = torch.cuda.CUDAGraph()
        captured = True
        try:
            with torch.cuda.graph(g, capture_error_mode=self.capture_error_mode):
                self.fwd_bwd()
                self.bucket.all_reduce()
                if os.environ.get("RWK_ABORT_CAPTURE_FOR_TEST") == "1":
                    # rehearsal of the fall-back: a capture that fails AFTER the collective was recorded
                    raise RuntimeError("capture aborted on purpose")
                self._opt_step()
        except Exception as exc:                         # refused by the runtime or the collective library
            self.capture_note = "all-reduce capture refused (%s: %s)" % (type(exc).__name__, str(exc)[:200])
            torch.cuda.synchronize()
            captured = False
        # the ranks agree before any of them replays the graph: its all-reduce would wait for ranks that have none
        if not self._all_ranks(captured):
            if captured:
                self.capture_note = "all-reduce capture refused on another rank; two graphs instead"
            del g
            self.loss = None
            return False
        # one replay, then the same sum done eagerly: every rank must hold identical (averaged) gradients, so the
        # spread of a checksum of the bucket over the ranks must be zero
        g.replay()
        torch.cuda.synchronize()
        import torch.distributed as dist
        probe = self.bucket.flat.double().sum().reshape(1)
        lo, hi = probe.clone(), probe.clone()
        if dist.is_available() and dist.is_initialized():
            dist.all_reduce(lo, op=dist.ReduceOp.MIN)
            dist.all_reduce(hi, op=dist.ReduceOp.MAX)
        if not bool(torch.isfinite(probe).all()) or float((hi - lo).abs()) > 1e-9 * max(1.0, float(hi.abs())):
            self.capture_note = "captured all-reduce left different gradients on the ranks; two graphs instead"
            self.loss = None
            return False
        self.g_fb, self.g_opt = g, None
        self._single_loss = self.loss = self.loss.detach()   # same storage, without the captured autograd graph
        self.collective_captured = True
        self.capture_note = "all-reduce captured inside the step graph"
        return True

    def step_graph(self):
        self.g_fb.replay()
        if self.world > 1 and not self.collective_captured:
            self._all_reduce()
            if self.opt_in_graph:
                self.g_opt.replay()
            else:
                self._opt_step()

    def run_steps(self, n: int):
        """``n`` training steps: whole multi-step replays first (``capture(steps_per_graph=k)``), then single ones.
        ``self.loss`` is the loss of the last step run; ``self.losses`` those of the last multi-step replay."""
        k = self.multi_k
        last_multi = False
        if k > 1:
            while n >= k:
                self.g_multi.replay()
                n -= k
                last_multi = True
        if n > 0:
            if self._single_loss is not None:
                self.loss = self._single_loss
            for _ in range(n):
                self.step_graph()
        elif last_multi:
            self.loss = self.losses[-1]

    def phase_times(self, steps: int = 30):
        """Median device time (ms) of the phases of the replayed step, from events on the launch stream, and the
        host time of a whole step: where a data-parallel step spends its time (forward+backward graph, gradient
        all-reduce, optimiser graph, and what the host adds between them)."""
        import time
        ev = [[torch.cuda.Event(enable_timing=True) for _ in range(4)] for _ in range(steps)]
        host = []
        two = self.world > 1 and not self.collective_captured
        for i in range(steps):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            ev[i][0].record()
            self.g_fb.replay()
            ev[i][1].record()
            if two:
                self._all_reduce()
                ev[i][2].record()
                if self.opt_in_graph:
                    self.g_opt.replay()
                else:
                    self._opt_step()
            ev[i][3].record()
            torch.cuda.synchronize()
            host.append((time.perf_counter() - t0) * 1e3)

        def med(v):
            v = sorted(v)
            return round(v[len(v) // 2], 4)
        out = {"step_device_ms": med([e[0].elapsed_time(e[3]) for e in ev]), "step_host_sync_ms": med(host)}
        if two:
            out["fwd_bwd_graph_ms"] = med([e[0].elapsed_time(e[1]) for e in ev])
            out["all_reduce_ms"] = med([e[1].elapsed_time(e[2]) for e in ev])
            out["optimizer_ms"] = med([e[2].elapsed_time(e[3]) for e in ev])
        return out


class EpochTrainer(Trainer):
    """The reference's epoch over its training batches (``rwgnn2/main.py:103-106``: ``for i in range(n_train_batches):
    train(epoch, adj_train[i], features_train[i], graph_indicator_train[i], y_train[i])``) as ONE replayed graph.

    ``batches``: the lists ``generate_batches`` returns, zipped - ``[(adj, features, graph_indicator, y), ...]``, all
    resident on the device (they are reused every epoch, ``main.py:70-72``).  ``epoch_eager()`` runs the steps one
    after the other; ``capture_epoch()`` records them once (after one eager epoch as warm-up, which trains like any
    other) and ``run_epoch()`` replays them: one replay start per epoch instead of one per step, no host work per
    batch.  ``epoch_losses`` holds one loss tensor per batch (read them after the epoch, as ``main.py:87`` does with
    ``loss.item()``); a learning-rate schedule (``main.py:80,108``) sets the rate between replays
    (``trainer.opt.set_lr``, the rate lives on the device).  Single process; data-parallel runs use ``Trainer``."""

    def __init__(self, model, batches, **kw):
        batches = [tuple(b) for b in batches]
        if not batches:
            raise ValueError("no batches")
        if kw.get("world", 1) != 1:
            raise ValueError("EpochTrainer is single-process; use Trainer for data-parallel steps")
        super().__init__(model, batches[0], **kw)
        self.batches = batches
        self.epoch_losses = []
        self.g_epoch = None

    def set_batch(self, batch_t):
        self.adj, self.feats, self.gi, self.y = batch_t

    def _epoch(self):
        losses = []
        for b in self.batches:
            self.set_batch(b)
            self.fwd_bwd()
            self._opt_step()
            losses.append(self.loss.detach())
        return losses

    def epoch_eager(self):
        self.epoch_losses = self._epoch()
        return self.epoch_losses

    def capture_epoch(self):
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s), stale_autograd_graph_is_an_error():
            self.epoch_eager()
        torch.cuda.current_stream().wait_stream(s)
        torch.cuda.synchronize()
        self.loss = None                  # drop the warm-up's autograd graph (see Trainer.capture)
        self.g_epoch = torch.cuda.CUDAGraph()
        with stale_autograd_graph_is_an_error():
            with torch.cuda.graph(self.g_epoch, capture_error_mode=self.capture_error_mode):
                self.epoch_losses = self._epoch()
        self.loss = self.epoch_losses[-1]

    def run_epoch(self):
        self.g_epoch.replay()
        return self.epoch_losses


class StreamTrainer(Trainer):
    """Training over batches that are never resident twice: a pass over a ``gnn_amd.data.DeviceDataset`` in batches
    of graph ids, each assembled on the device right before its step (the reference assembles all batches of a fold
    up front, ``rwgnn2/main.py:70-72``; a data set of a million graphs is streamed instead).

    Batch t+1 (``DeviceDataset.batch``: one upload, one launch, the structure included) is issued on a side stream
    before step t is launched, into one of ``depth`` preallocated slots, so its assembly runs under step t and
    nothing is allocated per batch; the side stream waits for the step that last used the slot, the launch stream
    waits for the assembly.  Steps are launched eagerly (every batch has its own shape, so there is no graph to
    replay); the host never waits for the device inside a pass.  Same arithmetic, same order as stepping the same
    batches one by one: the parameters after a pass are the same bits."""

    def __init__(self, model, dataset, max_graphs: int, max_nodes: int, max_nnz: int, depth: int = 3,
                 side_stream=None, **kw):
        if kw.get("world", 1) != 1 and kw.get("capture_collective"):
            raise ValueError("a streamed step is launched eagerly: nothing to capture the collective in")
        self.dataset = dataset
        self.slots = dataset.make_slots(depth, max_graphs, max_nodes, max_nnz)
        # side_stream=True: every assembly on a side stream, one batch ahead; False: on the step's own stream, in
        # front of the step; None: chosen per pass - the side stream where a step has long kernels for the assembly
        # to hide under (measured on stream_h64: 1.00 against 1.06 ms per step), the step's own stream where the step
        # is a chain of short launches that an assembly sharing the chip slows by as much as it takes itself, and
        # the cross-stream waits come on top (the one-call steps at hidden_dim <= 4; proteins1024: 0.104 against
        # 0.110 ms per step) - and there the pass call has no assembly launch at all where the batch allows it: the
        # forward's first launch reads the data set itself (rwk.h: RWK_PASS_IN_FORWARD)
        self.side_stream = side_stream
        self.build_stream = torch.cuda.Stream(device=dataset.device) if side_stream is not False else None
        self._side = self.build_stream          # the current pass's choice
        self.slot_free = [None] * depth         # event: the step that last read slot k has been queued and run
        first = dataset.batch(list(range(min(2, dataset.n_graphs))))
        super().__init__(model, first, **kw)
        self.pass_losses = []
        self.raw_steps = 0                      # steps that ran as one library call on a slot's buffers

    def _prepare(self, plan, i, k, raw: bool):
        """Issue the assembly of batch ``i`` of the plan into slot ``k`` (on the side stream when there is one);
        returns (batch tensors, or None when ``raw``: the step then reads the slot's buffers itself; ready event or
        None)."""
        from . import _lib
        ds = self.dataset
        side = self._side
        if side is not None and self.slot_free[k] is not None:
            side.wait_event(self.slot_free[k])
        stream = side.cuda_stream if side is not None else _lib.current_stream(ds.device)
        with _lib.device_guard(ds.device):
            ds.launch_planned(plan, i, self.slots[k], stream, lean=raw)
        ready = None
        if side is not None:
            ready = torch.cuda.Event()
            ready.record(side)
        return (None if raw else ds.wrap_slot(plan, i, self.slots[k])), ready

    def run_pass(self, id_lists, keep_losses: bool = True):
        """One training step per entry of ``id_lists`` (lists of graph ids of the data set), in order.  Returns the
        per-batch loss tensors (read them after the pass)."""
        main = torch.cuda.current_stream(self.dataset.device)
        losses = []
        if not len(id_lists):
            return losses
        plan = self.dataset.plan(id_lists)       # shapes on the host, offset tables in one upload for the whole pass
        if self._train_step is None and self.tail is not None:
            from .composite import TrainStep
            self._train_step = TrainStep(self)
        # the whole step as one library call on the slot's buffers (no tensors per batch) where that applies
        raw = self._train_step is not None and self.world == 1 and \
            self._train_step.slots_supported(self.model, self.dataset)
        from . import composite
        self._side = self.build_stream if (self.side_stream or (self.side_stream is None and not raw)) else None
        n, t = len(id_lists), 0
        while t < n:
            if not (raw and composite.ONE_CALL_PASS):
                self._run_range(plan, t, n, raw, main, losses, keep_losses)
                break
            # every step of the pass (and the assembly in front of it) from ONE library call
            side = self._side
            self.loss = None
            k, ls = self._train_step.run_pass_slots(
                self.model, self.dataset, self.slots, plan, t, main.cuda_stream,
                side.cuda_stream if side is not None else None,
                in_forward=self.side_stream is None and composite.ASSEMBLE_IN_FORWARD)
            if k:
                self.raw_steps += k
                self.slot_free = [None] * len(self.slots)    # the next assembly waits for the whole launch stream
                if keep_losses:
                    losses.extend(ls.unbind(0))
                t += k
            if t < n:
                # a batch the pass call does not serve (or a change of the last launch's shape): the per-batch way
                self._run_range(plan, t, t + 1, raw, main, losses, keep_losses)
                t += 1
        self.pass_losses = losses
        return losses

    def evaluate(self, id_lists):
        """The model's output (log-probabilities, one tensor per entry of ``id_lists``) for batches of the data set, in
        the model's current mode - the validation / test loop of ``rwgnn2/main.py:110-117,148-151`` (call
        ``model.eval()`` first, as the reference does).  In evaluation mode a batch the one-launch forward serves is
        not assembled at all (``gnn_amd.eval.eval_forward_planned``: its workgroups read the graphs from the data
        set); any other batch is assembled into a slot and goes through ``model(...)``.  No gradients are recorded."""
        from .eval import eval_forward_planned
        outs = []
        if not len(id_lists):
            return outs
        plan = self.dataset.plan(id_lists)
        main = torch.cuda.current_stream(self.dataset.device)
        depth = len(self.slots)
        # every slot may still be read by queued steps or be written on the side stream: both end before this
        if self.build_stream is not None:
            main.wait_stream(self.build_stream)
        used = 0
        with torch.no_grad():
            for i in range(len(id_lists)):
                out = None if self.model.training else eval_forward_planned(self.model, self.dataset, plan, i)
                if out is None:
                    batch = self.dataset.batch_planned(plan, i, self.slots[used % depth])
                    used += 1
                    out = self.model(batch[0], batch[1], batch[2])
                outs.append(out)
        self.slot_free = [None] * depth
        return outs

    def _run_range(self, plan, t_begin: int, t_end: int, raw: bool, main, losses, keep_losses: bool):
        """Batches ``t_begin .. t_end - 1`` of the plan, one step each, launched batch by batch."""
        depth = len(self.slots)
        if self._side is not None:
            self._side.wait_stream(main)
        else:
            self.slot_free = [None] * depth      # (events of an earlier pass on the side stream: nothing waits there now)
        side = self._side is not None
        nxt = self._prepare(plan, t_begin, t_begin % depth, raw)
        for t in range(t_begin, t_end):
            batch, ready = nxt
            if side and t + 1 < t_end:
                nxt = self._prepare(plan, t + 1, (t + 1) % depth, raw)
            if ready is not None:
                main.wait_event(ready)
            self.loss = None
            if not (raw and self._train_step.run_slot(self.model, self.dataset, self.slots[t % depth],
                                                      plan.metas[t])):
                if batch is None:
                    # (rare: a batch the one-call step does not serve) the lean assembly left the tensors' own
                    # arrays unwritten: assemble again, in full, in front of the step
                    from . import _lib
                    self.dataset.launch_planned(plan, t, self.slots[t % depth],
                                                _lib.current_stream(self.dataset.device))
                    batch = self.dataset.wrap_slot(plan, t, self.slots[t % depth])
                self.adj, self.feats, self.gi, self.y = batch
                self.fwd_bwd()
                if self.world > 1:
                    self._all_reduce()
            else:
                self.raw_steps += 1
            self._opt_step()
            if side:
                done = torch.cuda.Event()
                done.record(main)
                self.slot_free[t % depth] = done
            elif t + 1 < t_end:
                # same stream: the next assembly is ordered behind this step by the stream itself
                nxt = self._prepare(plan, t + 1, (t + 1) % depth, raw)
            if keep_losses:
                losses.append(self.loss.detach())
        self.pass_losses = losses
        return losses
