"""``torch.autograd.Function`` around the HIP kernels of the random-walk-kernel layer.

Replaces ``rwgnn2/model.py:43-67`` (input projection + sigmoid, the ``max_step``-step propagation on the
batch graph and on the hidden graphs, the (H*S) x N contraction, dropout, product, per-graph segment
sum, normalisation and concatenation) and their autograd backward.

Two entry modes:
* narrow inputs (``input_dim <= 16``, e.g. the node degree): the projection ``features @ W^T + b`` and its
  weight/bias gradients run in our own streaming kernels (a GEMM library spends ~0.7 ms on the N-long
  reduction of a 4 x N by N x 1 product);
* wide inputs: the caller applies ``fc`` with torch (a real GEMM) and passes the pre-activation.

All buffers are torch tensors; the library itself allocates nothing.
"""
from __future__ import annotations

import torch

from . import _lib
from .structure import GraphStructure, use_tile_kernel

__all__ = ["rw_kernel_layer", "rw_kernel_layer_projected", "choose_stream_len", "FUSED_PROJECTION_MAX_INPUT_DIM"]

FUSED_PROJECTION_MAX_INPUT_DIM = 16
#: launch the hidden-graph kernels together with the projection kernels they are independent of (one launch
#: instead of two, forward and backward); False keeps the stand-alone launches (tests compare the two)
MERGED_LAUNCHES = True
#: do the last transposed propagation step inside the projection's backward launch, the hidden-graph backward
#: beside the projection-gradient sum (hidden_dim 4, narrow inputs, at least two steps, a matrix the tile SpMM
#: kernel serves), and the batch statistics of the head inside the K reduction; False keeps the stand-alone
#: launches (tests compare the two)
FUSED_STEP_LAUNCHES = True
#: single-process training step with the optimiser's update in the last launch (optim.StepTail): leave out the head's
#: finishing launch as well - the batch-norm backward moves into the fused backward kernel (each lane forms the
#: gradient elements of the graphs it walks), the head's parameter-gradient sums into the last launch; False keeps
#: the head's own launch (tests compare the two)
FOLD_HEAD_FINISH = True
# matrix-core path: the forward's dropout keep bits travel to the backward in memory (8 bytes per (node, hidden
# graph)) instead of being drawn a second time
KEEP_BITS_HANDOVER = True
#: without dropout (evaluation mode, dropout 0) and hidden_dim <= 4 the layer runs through per-graph Gram matrices
#: (csrc/rwk_gram.hip): 16 multiply-adds per node and step instead of 2*H*S*D; False keeps the dropout-shaped kernels
#: with the identity mask (tests compare the two)
GRAM_FAST_PATH = True
#: block-diagonal batches (sorted graph ids, no stored entry between two graphs, graphs that fit the LDS): projection,
#: every propagation step and the hidden graphs in ONE launch, one workgroup per graph (csrc/rwk_eval.hip), instead of
#: the projection launch followed by one SpMM launch per step; False keeps those (tests compare the two)
GRAPH_X_CHAIN = True
#: block-diagonal batches of large graphs (more than GRAPH_CHAIN_MIN_NODES nodes, or rows of more than 64 stored
#: entries): the propagation chain and its backward per graph in LDS (csrc/rwk_chain.hip); False keeps the SpMM launches
GRAPH_CHAIN = True
GRAPH_CHAIN_MIN_NODES = 1024
#: ... and enough of them to give every CU a graph: a workgroup walks its graph on ONE CU (a step of a 3000-node graph
#: takes 6-10 us there, LDS-bound), so 64 such graphs are served as fast by SpMM launches that spread every step over
#: the whole chip (38 against 41 us for four steps), 256 of them faster by the chain (46 against 61 us)
GRAPH_CHAIN_MIN_GRAPHS = 128
#: graphs beyond this many nodes on average stay on the stream kernels (a Gram wavefront walks its graph alone)
GRAM_MAX_MEAN_NODES = 4096

def choose_stream_len(n_nodes: int, hidden_dim: int, n_hidden: int, size_hidden: int, max_step: int,
                      backward: bool = False) -> int:
    """Sorted positions walked sequentially by one lane group of the fused forward (or backward) kernel:
    the library's own rule, which knows how many workgroups of each kernel a CU keeps resident."""
    return int(_lib.load().rwk_fused_stream_len(n_nodes, hidden_dim, n_hidden, size_hidden, max_step,
                                                int(bool(backward))))


def _plan(lib, st: GraphStructure, d: int, n_hidden: int, s: int, steps: int):
    """Launch constants of one (batch structure, hyper-parameters) pair, computed once: stream lengths of the fused
    kernels, their partial-sum buffer sizes, workspace size.  Batches are reused every epoch, so a training loop
    finds them in the structure's cache from the second step on (seven host calls into the library less per step)."""
    key = (d, n_hidden, s, steps)
    plan = st.plans.get(key)
    if plan is None:
        n, g = st.n, st.n_graphs
        sl_f = choose_stream_len(n, d, n_hidden, s, steps)
        sl_b = choose_stream_len(n, d, n_hidden, s, steps, backward=True)
        n_part = int(lib.rwk_fused_dz_partials_count(n, sl_b, d, n_hidden, s, steps))
        plan = {
            "stream_len": sl_f, "stream_len_b": sl_b,
            "k_slots": int(lib.rwk_fused_k_partials_size(n, sl_f, g, n_hidden, steps)),
            "n_part": n_part,
            "dz_rows32": int(lib.rwk_reduce_rows_level1(n_part)) if n_part > 32 else n_part,
            "ws": int(lib.rwk_fused_backward_workspace(n, d, n_hidden, s, steps)),
            "stats_size": int(lib.rwk_head_stats_size(g, steps * n_hidden)),
            "stats_blocks": int(lib.rwk_fused_reduce_stats_blocks(g, steps * n_hidden)),
            "spb_blocks": int(lib.rwk_spmm_project_backward_blocks(n)),
        }
        st.plans[key] = plan
    return plan


def _reduce_rows(lib, partials2d, stream):
    """Deterministic column sum of a (P, V) tensor."""
    p, v = partials2d.shape
    out = torch.empty(v, dtype=torch.float32, device=partials2d.device)
    ws_n = lib.rwk_reduce_rows_workspace(p, v)
    ws = torch.empty(max(ws_n, 1), dtype=torch.float32, device=partials2d.device)
    _lib.check(lib.rwk_reduce_rows(partials2d.data_ptr(), p, v, out.data_ptr(), ws.data_ptr(), stream),
               "rwk_reduce_rows")
    return out


class _RWKernelLayer(torch.autograd.Function):
    """inp is the pre-activation (N, D) when fc_weight is None, else the raw features (N, F)."""

    @staticmethod
    def forward(ctx, inp, fc_weight, fc_bias, adj_hidden, features_hidden, st: GraphStructure, max_step: int,
                normalize: bool, drop_p: float, rng_state, handoff=None):
        lib = _lib.load()
        if not inp.is_cuda:
            raise RuntimeError("gnn_amd runs on HIP devices only (no CPU fallback)")
        dev = inp.device
        n_hidden, s, d = features_hidden.shape
        n, g, steps = st.n, st.n_graphs, int(max_step)
        projected = fc_weight is not None
        if inp.shape[0] != n or (not projected and inp.shape[1] != d):
            raise ValueError(f"input has shape {tuple(inp.shape)} for {n} nodes and hidden_dim {d}")
        if n_hidden > 64 or d > 64 or steps > 4:
            raise NotImplementedError(
                "one launch covers hidden_graphs <= 64, hidden_dim <= 64, max_step <= 4 "
                f"(got {n_hidden}, {d}, {steps}): call rw_kernel_layer, which composes larger shapes")
        drop = float(drop_p) if (drop_p and rng_state is not None) else 0.0
        with _lib.device_guard(dev):
            stream = _lib.current_stream(dev)
            adj_hidden_c = adj_hidden.detach().contiguous()
            feat_c = features_hidden.detach().contiguous()
            xs = torch.empty((steps, n, d), dtype=torch.float32, device=dev)
            inp_c = inp.detach().contiguous()
            w_c = None
            z = torch.empty((steps, n_hidden, s, d), dtype=torch.float32, device=dev)
            ah = torch.empty((n_hidden, s, s), dtype=torch.float32, device=dev)
            rng_used = torch.empty(2, dtype=torch.int64, device=dev) if drop else None
            rng_in = _lib.ptr(rng_state) if drop else None
            f_in = inp_c.shape[1]
            fuse_step = (projected and d == 4 and steps >= 2 and MERGED_LAUNCHES and FUSED_STEP_LAUNCHES
                         and 4 * f_in + 4 <= 64 and st.nnz > 0
                         and use_tile_kernel(n, st.nnz, st.csr_t.max_degree, 4))
            x_chain = big_chain = chain_bwd = False
            max_nodes = 0
            if (projected and d == 4 and MERGED_LAUNCHES and GRAPH_X_CHAIN and st.perm is None and st.nnz > 0
                    and st.csr.col is not None):
                from .eval import _facts
                max_nodes, block_diagonal = _facts(st)
                # graphs of thousands of nodes and rows of hundreds of entries: a workgroup of 1024 threads per graph,
                # hub rows by a wavefront each (rwk_chain.hip); small sparse graphs: 256 threads, a thread per row
                big = max_nodes > GRAPH_CHAIN_MIN_NODES or max(st.csr.max_degree, st.csr_t.max_degree) > 64
                if block_diagonal and big and GRAPH_CHAIN and g >= GRAPH_CHAIN_MIN_GRAPHS:
                    big_chain = bool(lib.rwk_graph_chain_forward_lds(f_in, max_nodes, s))
                    chain_bwd = bool(fuse_step and st.csr_t.col is not None
                                     and lib.rwk_graph_chain_backward_lds(f_in, max_nodes))
                elif block_diagonal and st.csr.max_degree <= 64:
                    # (rows of more than 64 stored entries are summed in tree order by the tile SpMM: left to it)
                    x_chain = bool(lib.rwk_x_chain_lds(f_in, max_nodes, s))
            if big_chain:                                                        # model.py:39-43,55-56 in one launch
                w_c = fc_weight.detach().contiguous()
                csr = st.csr
                _lib.check(lib.rwk_graph_chain_forward(
                    inp_c.data_ptr(), w_c.data_ptr(), fc_bias.detach().contiguous().data_ptr(), n, f_in,
                    csr.crow.data_ptr(), _lib.ptr(csr.col), None if csr.unit_values else _lib.ptr(csr.val),
                    st.seg_ptr.data_ptr(), g, max_nodes, steps, xs.data_ptr(), _lib.ptr(adj_hidden_c),
                    feat_c.data_ptr(), n_hidden, s, ah.data_ptr(), z.data_ptr(), rng_in, _lib.ptr(rng_used), stream),
                    "rwk_graph_chain_forward")
                x_chain = True                                                   # no SpMM launches below
            elif x_chain:                                                        # the same for small sparse graphs
                w_c = fc_weight.detach().contiguous()
                csr = st.csr
                _lib.check(lib.rwk_x_chain_hidden_forward(
                    inp_c.data_ptr(), w_c.data_ptr(), fc_bias.detach().contiguous().data_ptr(), n, f_in,
                    csr.crow.data_ptr(), _lib.ptr(csr.col), None if csr.unit_values else _lib.ptr(csr.val),
                    st.seg_ptr.data_ptr(), g, max_nodes, steps, xs.data_ptr(), _lib.ptr(adj_hidden_c),
                    feat_c.data_ptr(), n_hidden, s, ah.data_ptr(), z.data_ptr(), rng_in, _lib.ptr(rng_used), stream),
                    "rwk_x_chain_hidden_forward")
            elif projected and d == 4 and MERGED_LAUNCHES:                     # model.py:39-43,56 in one launch
                w_c = fc_weight.detach().contiguous()
                _lib.check(lib.rwk_project_hidden_forward(
                    inp_c.data_ptr(), w_c.data_ptr(), fc_bias.detach().contiguous().data_ptr(), n, inp_c.shape[1], d,
                    xs.data_ptr(), _lib.ptr(adj_hidden_c), feat_c.data_ptr(), n_hidden, s, steps, ah.data_ptr(),
                    z.data_ptr(), rng_in, _lib.ptr(rng_used), stream), "rwk_project_hidden_forward")
            else:
                if projected:                                                    # model.py:43
                    w_c = fc_weight.detach().contiguous()
                    _lib.check(lib.rwk_project_forward(inp_c.data_ptr(), w_c.data_ptr(),
                                                       fc_bias.detach().contiguous().data_ptr(), n, inp_c.shape[1],
                                                       d, xs.data_ptr(), stream), "rwk_project_forward")
                else:
                    torch.sigmoid(inp_c, out=xs[0])
                _lib.check(lib.rwk_hidden_forward(_lib.ptr(adj_hidden_c), feat_c.data_ptr(), n_hidden, s, d, steps,
                                                  ah.data_ptr(), z.data_ptr(), rng_in, _lib.ptr(rng_used), stream),
                           "rwk_hidden_forward")
            for i in range(1, 1 if x_chain else steps):
                st.spmm(xs[i - 1], xs[i])                                        # model.py:55
            use_gram = (GRAM_FAST_PATH and drop == 0.0 and n <= GRAM_MAX_MEAN_NODES * g
                        and bool(lib.rwk_gram_supported(d, n_hidden, s, steps)))
            if use_gram:                                                         # model.py:45-64 without a mask
                k = torch.empty((g, steps * n_hidden), dtype=torch.float32, device=dev)
                gram_c = torch.empty((g, steps, 16), dtype=torch.float32, device=dev) \
                    if any(ctx.needs_input_grad) else None
                _lib.check(lib.rwk_gram_forward(xs.data_ptr(), z.data_ptr(), st.seg_ptr.data_ptr(), _lib.ptr(st.perm),
                                                st.counts_f32.data_ptr(), n, g, d, n_hidden, s, steps,
                                                int(bool(normalize)), k.data_ptr(), _lib.ptr(gram_c), stream),
                           "rwk_gram_forward")
                if handoff is not None:
                    handoff["fold_ok"] = False
                ctx.st = st
                ctx.cfg = (steps, bool(normalize), drop, projected)
                ctx.fuse_step = fuse_step
                ctx.chain_bwd = (chain_bwd, max_nodes)
                ctx.handoff = handoff
                ctx.gram = True
                ctx.save_for_backward(xs, z, ah, adj_hidden_c, rng_used, inp_c if projected else None, w_c, gram_c)
                return k
            plan = _plan(lib, st, d, n_hidden, s, steps)
            stream_len = plan["stream_len"]
            k_partials = torch.empty(plan["k_slots"], dtype=torch.float32, device=dev)
            # matrix-core path, training with dropout: the forward leaves its keep bits for the backward
            keep = None
            if drop and KEEP_BITS_HANDOVER and any(ctx.needs_input_grad):
                words = lib.rwk_fused_keep_words(n, d, n_hidden, s, steps)
                if words > 0:
                    keep = torch.empty(words, dtype=torch.int64, device=dev)
            _lib.check(lib.rwk_fused_forward_bits(xs.data_ptr(), z.data_ptr(), st.gid.data_ptr(), _lib.ptr(st.perm),
                                                  n, d, n_hidden, s, steps, stream_len, drop, _lib.ptr(rng_used),
                                                  k_partials.data_ptr(), _lib.ptr(keep), stream),
                       "rwk_fused_forward_bits")
            k = torch.empty((g, steps * n_hidden), dtype=torch.float32, device=dev)
            if handoff is not None and FUSED_STEP_LAUNCHES and steps * n_hidden <= 256:
                # the head's batch statistics (and its draw of the generator state) ride in the K reduction
                head_drop = bool(handoff.get("head_drop")) and rng_state is not None
                stats = torch.empty(plan["stats_size"], dtype=torch.float32, device=dev)
                rng_head = torch.empty(2, dtype=torch.int64, device=dev) if head_drop else None
                _lib.check(lib.rwk_fused_reduce_stats(
                    k_partials.data_ptr(), st.seg_ptr.data_ptr(), st.counts_f32.data_ptr(), g, n_hidden, steps,
                    stream_len, drop, int(bool(normalize)), k.data_ptr(), stats.data_ptr(),
                    _lib.ptr(rng_state) if head_drop else None, _lib.ptr(rng_head), stream), "rwk_fused_reduce_stats")
                handoff["stats"] = stats
                handoff["n_stats"] = plan["stats_blocks"]
                handoff["rng_used"] = rng_head
            else:
                _lib.check(lib.rwk_fused_reduce(k_partials.data_ptr(), st.seg_ptr.data_ptr(),
                                                st.counts_f32.data_ptr(), g, n_hidden, steps, stream_len, drop,
                                                int(bool(normalize)), k.data_ptr(), stream), "rwk_fused_reduce")
        if handoff is not None:
            tail = handoff.get("tail")
            handoff["fold_ok"] = bool(
                FOLD_HEAD_FINISH and fuse_step and tail is not None and s > 1 and not ctx.needs_input_grad[0]
                and "stats" in handoff and lib.rwk_fused_backward_bn_supported(d, n_hidden, s, steps))
        ctx.st = st
        ctx.cfg = (steps, bool(normalize), drop, projected)
        ctx.fuse_step = fuse_step
        ctx.chain_bwd = (chain_bwd, max_nodes)
        ctx.handoff = handoff
        ctx.gram = False
        ctx.save_for_backward(xs, z, ah, adj_hidden_c, rng_used, inp_c if projected else None, w_c, keep)
        return k

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, d_k):
        lib = _lib.load()
        xs, z, ah, adj_hidden_c, rng_used, feats, w_c, keep = ctx.saved_tensors
        st: GraphStructure = ctx.st
        steps, normalize, drop, projected = ctx.cfg
        _, n_hidden, s, d = z.shape
        n = st.n
        plan = _plan(lib, st, d, n_hidden, s, steps)
        stream_len = plan["stream_len_b"]
        dev = xs.device
        with _lib.device_guard(dev):
            stream = _lib.current_stream(dev)
            d_k = d_k.contiguous().float()
            dx = torch.empty_like(xs)
            fold = ctx.handoff.pop("head_fold", None) if ctx.handoff is not None else None
            if ctx.gram:
                # Gram path: `keep` is C (G, L, 4, 4); dX from dC = dK M, dZ through dM = sum_g dK C
                dz_partials = torch.empty((1, steps * n_hidden * s * d), dtype=torch.float32, device=dev)
                _lib.check(lib.rwk_gram_backward(xs.data_ptr(), z.data_ptr(), st.seg_ptr.data_ptr(), _lib.ptr(st.perm),
                                                 st.counts_f32.data_ptr(), d_k.data_ptr(), keep.data_ptr(), n,
                                                 st.n_graphs, d, n_hidden, s, steps, int(normalize), dx.data_ptr(),
                                                 dz_partials.data_ptr(), stream), "rwk_gram_backward")
                n_part = 1
            else:
                n_part = plan["n_part"]
                if n_part <= 0:
                    raise NotImplementedError("unsupported hyper-parameters for the fused backward kernel")
                dz_partials = torch.empty((n_part, steps * n_hidden * s * d), dtype=torch.float32, device=dev)
                ws_n = plan["ws"]
                ws = torch.empty(ws_n, dtype=torch.float32, device=dev) if ws_n else None
            if ctx.gram:
                pass
            elif fold is not None:
                # the head left its finishing launch out: d_k is a placeholder, the kernel forms it (head.py)
                _lib.check(lib.rwk_fused_backward_bn(
                    xs.data_ptr(), z.data_ptr(), st.gid.data_ptr(), _lib.ptr(st.perm), st.counts_f32.data_ptr(), n, d,
                    n_hidden, s, steps, stream_len, drop, int(normalize), _lib.ptr(rng_used), dx.data_ptr(),
                    dz_partials.data_ptr(), fold["k"].data_ptr(), fold["dy"].data_ptr(), fold["mean"].data_ptr(),
                    fold["invstd"].data_ptr(), fold["bn_w"].data_ptr(), fold["partials"].data_ptr(),
                    fold["partials"].shape[0], fold["partials"].shape[1], fold["o_gamma"], fold["d_loss"].data_ptr(),
                    st.n_graphs, stream), "rwk_fused_backward_bn")
            else:
                _lib.check(lib.rwk_fused_backward_bits(xs.data_ptr(), z.data_ptr(), st.gid.data_ptr(),
                                                       _lib.ptr(st.perm), d_k.data_ptr(), st.counts_f32.data_ptr(), n, d,
                                                       n_hidden, s, steps, stream_len, drop, int(normalize),
                                                       _lib.ptr(rng_used), dx.data_ptr(), dz_partials.data_ptr(),
                                                       _lib.ptr(ws), _lib.ptr(keep), stream),
                           "rwk_fused_backward_bits")
            # dZ partials: first level here (to at most 32 rows), the rest inside rwk_hidden_backward.  It is
            # independent of the SpMM chain and rides in the first SpMM's launch when there is one.
            dz, dz_rows, job = dz_partials, n_part, None
            if n_part > 32:
                dz_rows = plan["dz_rows32"]
                dz = torch.empty((dz_rows, dz_partials.shape[1]), dtype=torch.float32, device=dev)
                job = (dz_partials, dz)
            chain_bwd, max_nodes = ctx.chain_bwd
            last = 1 if ctx.fuse_step else 0                                    # the fused launch does step 1 -> 0
            for i in range(steps - 1, steps - 1 if chain_bwd else last, -1):    # dX[i-1] += A^T dX[i]
                st.spmm(dx[i], dx[i - 1], transpose=True, accumulate=True,
                        reduce_job=job if MERGED_LAUNCHES else None)
                if MERGED_LAUNCHES:
                    job = None
            d_adj = torch.empty_like(adj_hidden_c)
            d_feat = torch.empty((n_hidden, s, d), dtype=torch.float32, device=dev)
            x0 = xs[0]
            d_w = d_b = None
            if ctx.fuse_step:
                f = feats.shape[1]
                csr_t = st.csr_t
                blocks = st.n_graphs if chain_bwd else plan["spb_blocks"]
                partials = torch.empty((blocks, d * f + d), dtype=torch.float32, device=dev)
                d_inp = torch.empty_like(feats) if ctx.needs_input_grad[0] else None
                red_in, red_out = job if job is not None else (None, None)
                if chain_bwd:       # the whole transposed chain and the projection's backward, per graph in LDS
                    _lib.check(lib.rwk_graph_chain_backward(
                        dx.data_ptr(), x0.data_ptr(), feats.data_ptr(), w_c.data_ptr(), n, f, csr_t.crow.data_ptr(),
                        _lib.ptr(csr_t.col), None if csr_t.unit_values else _lib.ptr(csr_t.val),
                        st.seg_ptr.data_ptr(), st.n_graphs, max_nodes, steps, partials.data_ptr(), _lib.ptr(d_inp),
                        _lib.ptr(red_in), red_in.shape[0] if job else 0, red_in.shape[1] if job else 0,
                        _lib.ptr(red_out), stream), "rwk_graph_chain_backward")
                else:
                    _lib.check(lib.rwk_spmm_project_backward(
                        dx[0].data_ptr(), dx[1].data_ptr(), x0.data_ptr(), feats.data_ptr(), w_c.data_ptr(), n, f,
                        csr_t.crow.data_ptr(), _lib.ptr(csr_t.col), _lib.ptr(csr_t.col16),
                        None if (csr_t.col16 is not None and csr_t.unit_values) else _lib.ptr(csr_t.val),
                        partials.data_ptr(), _lib.ptr(d_inp), _lib.ptr(red_in), red_in.shape[0] if job else 0,
                        red_in.shape[1] if job else 0, _lib.ptr(red_out), stream), "rwk_spmm_project_backward")
                wb = torch.empty(d * f + d, dtype=torch.float32, device=dev)
                tail = ctx.handoff.get("tail") if ctx.handoff is not None else None
                head = ctx.handoff.pop("head_grads", None) if tail is not None else None
                if tail is not None and head is not None and s > 1 and not ctx.needs_input_grad[0]:
                    # single-process training step: the optimiser's update rides in this launch (optim.StepTail)
                    head_grads, hc, hp, hnc = head
                    tail.launch(lib, stream, tail.head_sources(head_grads, hc, hp, hnc), _lib.ptr(adj_hidden_c),
                                ah.data_ptr(), z.data_ptr(), dz.data_ptr(), dz_rows, n_hidden, s, d, steps,
                                _lib.ptr(d_adj), d_feat.data_ptr(), partials.data_ptr(), blocks, d * f + d,
                                wb.data_ptr(), head_fold=fold, n_graphs=st.n_graphs,
                                head_columns=tail.head_columns(hc, hp, hnc) if fold is not None else None)
                    ctx.handoff["keep"] = (head_grads, fold)  # the launch reads them asynchronously
                else:
                    if fold is not None:
                        raise RuntimeError("the head left its finishing launch to a step tail that did not run")
                    _lib.check(lib.rwk_hidden_backward_reduce(
                        _lib.ptr(adj_hidden_c), ah.data_ptr(), z.data_ptr(), dz.data_ptr(), dz_rows, n_hidden, s, d,
                        steps, _lib.ptr(d_adj), d_feat.data_ptr(), partials.data_ptr(), blocks, d * f + d,
                        wb.data_ptr(), stream), "rwk_hidden_backward_reduce")
                d_w = wb[:d * f].view(d, f)
                d_b = wb[d * f:]
                return d_inp, d_w, d_b, d_adj, d_feat, None, None, None, None, None, None
            if job is not None:
                _lib.check(lib.rwk_reduce_rows_to32(dz_partials.data_ptr(), n_part, dz_partials.shape[1],
                                                    dz.data_ptr(), stream), "rwk_reduce_rows_to32")
            merged = projected and MERGED_LAUNCHES
            if not merged:
                _lib.check(lib.rwk_hidden_backward(_lib.ptr(adj_hidden_c), ah.data_ptr(), z.data_ptr(),
                                                   dz.data_ptr(), dz_rows, n_hidden, s, d, steps,
                                                   _lib.ptr(d_adj), d_feat.data_ptr(), stream),
                           "rwk_hidden_backward")
            if projected:
                f = feats.shape[1]
                blocks = lib.rwk_project_backward_blocks(n, d)
                partials = torch.empty((blocks, d * f + d), dtype=torch.float32, device=dev)
                d_inp = torch.empty_like(feats) if ctx.needs_input_grad[0] else None
                if merged:                                                       # both backward kernels, one launch
                    _lib.check(lib.rwk_project_hidden_backward(
                        dx.data_ptr(), x0.data_ptr(), feats.data_ptr(), w_c.data_ptr(), n, f, d, partials.data_ptr(),
                        _lib.ptr(d_inp), _lib.ptr(adj_hidden_c), ah.data_ptr(), z.data_ptr(), dz.data_ptr(), dz_rows,
                        n_hidden, s, steps, _lib.ptr(d_adj), d_feat.data_ptr(), stream),
                        "rwk_project_hidden_backward")
                else:
                    _lib.check(lib.rwk_project_backward(dx.data_ptr(), x0.data_ptr(), feats.data_ptr(),
                                                        w_c.data_ptr(), n, f, d, partials.data_ptr(),
                                                        _lib.ptr(d_inp), stream), "rwk_project_backward")
                wb = _reduce_rows(lib, partials, stream)
                d_w = wb[:d * f].view(d, f)
                d_b = wb[d * f:]
            else:
                d_inp = dx[0] * x0 * (1.0 - x0)                                  # sigmoid'
        # with a single step adj_hidden never enters the forward pass: the reference leaves its
        # .grad at None, and so do we
        return d_inp, d_w, d_b, (d_adj if steps > 1 else None), d_feat, None, None, None, None, None, None


def _composed(pre, adj_hidden, features_hidden, structure, max_step, normalize, drop_p, rng_state):
    from .wide import rw_kernel_layer_composed
    return rw_kernel_layer_composed(pre, adj_hidden, features_hidden, structure, max_step, normalize, drop_p, rng_state)


def _beyond_direct_range(features_hidden, max_step) -> bool:
    """Hyper-parameters the fused kernels do not take in one launch (the reference accepts any, main.py:22-27)."""
    n_hidden, _, d = features_hidden.shape
    return n_hidden > 64 or d > 64 or int(max_step) > 4


def rw_kernel_layer(pre, adj_hidden, features_hidden, structure: GraphStructure, max_step: int,
                    normalize: bool, drop_p: float = 0.0, rng_state=None, handoff=None):
    """(G, max_step*H) random-walk-kernel features; ``pre`` is ``fc(features)`` BEFORE the sigmoid.

    ``handoff``: a dict ``{"head_drop": bool}`` from a caller that feeds K to the fused head in training mode: the
    K reduction then also leaves the head's batch-statistics partials (``handoff["stats"]``, ``["n_stats"]``) and
    draws the head's dropout generator state (``["rng_used"]``), which saves the head its statistics launch."""
    if _beyond_direct_range(features_hidden, max_step):
        return _composed(pre, adj_hidden, features_hidden, structure, max_step, normalize, drop_p, rng_state)
    return _RWKernelLayer.apply(pre, None, None, adj_hidden, features_hidden, structure, max_step, normalize,
                                drop_p, rng_state, handoff)


def rw_kernel_layer_projected(features, fc_weight, fc_bias, adj_hidden, features_hidden,
                              structure: GraphStructure, max_step: int, normalize: bool, drop_p: float = 0.0,
                              rng_state=None, handoff=None):
    """Same, with the input projection ``sigmoid(features @ fc_weight.T + fc_bias)`` fused in
    (``features.shape[1] <= FUSED_PROJECTION_MAX_INPUT_DIM``)."""
    if _beyond_direct_range(features_hidden, max_step):
        pre = torch.nn.functional.linear(features, fc_weight, fc_bias)
        return _composed(pre, adj_hidden, features_hidden, structure, max_step, normalize, drop_p, rng_state)
    return _RWKernelLayer.apply(features, fc_weight, fc_bias, adj_hidden, features_hidden, structure, max_step,
                                normalize, drop_p, rng_state, handoff)
