"""Drop-in ``RW_NN``: same constructor, ``forward(adj, features, graph_indicator)`` and
``state_dict`` keys as ``rwgnn2/model.py:8-72`` (identical to ``rwgnn/model.py``), so that
``from gnn_amd import RW_NN`` can replace ``from model import RW_NN`` in ``rwgnn2/main.py:12``.

Lines :32-67 of the reference forward run in hand-written HIP kernels (``gnn_amd/csrc``).  The input
projection ``fc`` and the head (:68-72: batch-norm, two linears, log-softmax) stay ordinary torch modules
in the ``state_dict``; the head's arithmetic runs in two fused launches each way when its sizes allow
(``fused_head = False`` restores the torch operators).
"""
from __future__ import annotations

import torch
import torch.nn as nn
import torch.nn.functional as F
from torch.nn.parameter import Parameter

from .composite import model_step, model_step_supported
from .eval import eval_forward, eval_forward_supported
from .function import FUSED_PROJECTION_MAX_INPUT_DIM, rw_kernel_layer, rw_kernel_layer_projected
from .head import fused_head, fused_head_loss, head_supported
from .losses import cross_entropy
from .structure import get_structure

__all__ = ["RW_NN"]


class RW_NN(nn.Module):
    def __init__(self, input_dim, max_step, hidden_graphs, size_hidden_graphs, hidden_dim, penultimate_dim,
                 normalize, n_classes, dropout, device):
        super().__init__()
        self.max_step = max_step
        self.hidden_graphs = hidden_graphs
        self.size_hidden_graphs = size_hidden_graphs
        self.normalize = normalize
        self.device = device
        n_pairs = (size_hidden_graphs * (size_hidden_graphs - 1)) // 2
        # parameter names, shapes and creation order follow model.py:16-21 (same state_dict, same
        # consumption of the global RNG at construction)
        self.adj_hidden = Parameter(torch.empty(hidden_graphs, n_pairs))
        self.features_hidden = Parameter(torch.empty(hidden_graphs, size_hidden_graphs, hidden_dim))
        self.fc = nn.Linear(input_dim, hidden_dim)
        self.bn = nn.BatchNorm1d(hidden_graphs * max_step)
        self.fc1 = nn.Linear(hidden_graphs * max_step, penultimate_dim)
        self.fc2 = nn.Linear(penultimate_dim, n_classes)
        self.dropout = nn.Dropout(p=dropout)
        self.relu = nn.ReLU()
        self.sigmoid = nn.Sigmoid()
        # {seed, offset} of the counter-based dropout mask; lives on the device so that a captured
        # HIP graph advances it on replay.  Not part of the state_dict (the reference has no such key).
        self.register_buffer("_rng_state", torch.zeros(2, dtype=torch.int64), persistent=False)
        self._rng_seeded = False
        self.fused_head = True
        self.init_weights()

    def init_weights(self):
        """model.py:27-29."""
        self.adj_hidden.data.uniform_(-1, 1)
        self.features_hidden.data.uniform_(0, 1)

    def seed_dropout(self, seed: int, offset: int = 0):
        """Fix the dropout stream (otherwise seeded from torch's global CPU generator on first use)."""
        self._rng_state.copy_(torch.tensor([seed & 0x7FFFFFFFFFFFFFFF, offset], dtype=torch.int64))
        self._rng_seeded = True

    def forward(self, adj, features, graph_indicator):
        structure = get_structure(adj, graph_indicator)
        if not self.training and eval_forward_supported(self, structure, features):
            # model.eval(): the validation / test call of rwgnn2/main.py:110-117,148-151 as one launch
            return eval_forward(self, structure, features)
        return self._forward_launches(structure, features)

    def _forward_launches(self, structure, features):
        """The forward pass as separate launches (training mode, and every shape the one-launch evaluation forward
        does not serve)."""
        p = float(self._modules["dropout"].p)
        drop = p if self.training else 0.0
        if drop >= 1.0:
            # every product is dropped (model.py:58 with p = 1): K is exactly zero
            pre = self.fc(features)
            k = pre.new_zeros(structure.n_graphs, self.hidden_graphs * self.max_step)
            k = k + 0.0 * (pre.sum() + self.features_hidden.sum() + self.adj_hidden.sum())
            handoff = None
        else:
            handoff = self._head_handoff(structure, drop)
            if handoff is not None and "tail" not in handoff:
                entry = model_step_supported(self, structure, features, drop)
                if entry is not None:                  # the eager loop's step: one autograd node, one call each way
                    if drop > 0.0 and not self._rng_seeded:
                        self._seed_from_torch()
                    return model_step(self, structure, features, entry)
            k = self._rw(features, structure, drop, handoff)
        return self._head(k, drop, handoff)

    def _head_handoff(self, structure, drop):
        """Training mode with the fused head: ask the random-walk layer's K reduction for the head's batch
        statistics (one launch less), see :func:`gnn_amd.function.rw_kernel_layer`."""
        if not (self.training and self._head_fusable(structure.n_graphs, self.hidden_graphs * self.max_step, drop)):
            return None
        handoff = {"head_drop": drop > 0.0}
        tail = getattr(self, "_step_tail", None)
        if tail is not None and tail.armed:   # gnn_amd.optim.StepTail: the optimiser's update in the last backward launch
            handoff["tail"] = tail
        return handoff

    def _head(self, k, drop, handoff=None):
        if self._use_fused_head(k, drop):
            if drop > 0.0 and not self._rng_seeded:
                self._seed_from_torch()
            return fused_head(k, self.bn, self.fc1, self.fc2, drop, self.training,
                              self._rng_state if drop > 0.0 else None, handoff)
        out = self.bn(k)
        out = self.relu(self.fc1(out))
        out = self.dropout(out)
        out = self.fc2(out)
        return F.log_softmax(out, dim=1)

    def forward_with_loss(self, adj, features, graph_indicator, y):
        """``output = self(adj, features, graph_indicator); loss = F.cross_entropy(output, y)`` of the training loop
        (``rwgnn2/main.py:84-85``), returned as ``(output, loss)``.  With the fused head and gradients enabled the
        loss rides in the head's launches (:func:`gnn_amd.head.fused_head_loss`): its value is complete after
        ``loss.backward()``, and ``output`` is detached."""
        structure = get_structure(adj, graph_indicator)
        p = float(self.dropout.p)
        drop = p if self.training else 0.0
        if drop < 1.0 and torch.is_grad_enabled():
            handoff = self._head_handoff(structure, drop)
            k = self._rw(features, structure, drop, handoff)
            if self._use_fused_head(k, drop) and y.dtype == torch.int64 and y.dim() == 1:
                if drop > 0.0 and not self._rng_seeded:
                    self._seed_from_torch()
                return fused_head_loss(k, y, self.bn, self.fc1, self.fc2, drop, self.training,
                                       self._rng_state if drop > 0.0 else None, handoff)
            out = self._head(k, drop, handoff)
        else:
            out = self.forward(adj, features, graph_indicator)
        return out, cross_entropy(out, y)

    def _use_fused_head(self, k, drop):
        return k.is_cuda and self._head_fusable(k.shape[0], k.shape[1], drop)

    def _head_fusable(self, n_graphs, n_features, drop):
        mods = self._modules                   # plain dict reads (nn.Module.__getattr__ costs ~0.3 us each)
        bn, fc1, fc2 = mods["bn"], mods["fc1"], mods["fc2"]
        return (self.fused_head and drop < 1.0 and bn.affine and bn.track_running_stats
                and bn.momentum is not None and fc1._parameters["bias"] is not None
                and fc2._parameters["bias"] is not None and n_features == bn.num_features
                and head_supported(n_graphs, n_features, fc1.out_features, fc2.out_features))

    def _seed_from_torch(self):
        self.seed_dropout(int(torch.randint(0, 2 ** 62, (1,)).item()))

    def _rw(self, features, structure, drop, handoff=None):
        if drop > 0.0 and not self._rng_seeded:
            self._seed_from_torch()
        rng = self._rng_state if drop > 0.0 else None
        if features.shape[1] <= FUSED_PROJECTION_MAX_INPUT_DIM:
            return rw_kernel_layer_projected(features, self.fc.weight, self.fc.bias, self.adj_hidden,
                                             self.features_hidden, structure, self.max_step,
                                             bool(self.normalize), drop, rng, handoff)
        return rw_kernel_layer(self.fc(features), self.adj_hidden, self.features_hidden, structure,
                               self.max_step, bool(self.normalize), drop, rng, handoff)

    def rw_features(self, adj, features, graph_indicator):
        """The (G, max_step*H) input of ``bn`` (model.py:67), for tests and benchmarks."""
        structure = get_structure(adj, graph_indicator)
        drop = float(self.dropout.p) if self.training else 0.0
        return self._rw(features, structure, drop)
