"""TEST INFRASTRUCTURE ONLY: torch (CPU, any dtype) restatement of ``RW_GNN.forward`` up to the concatenated
per-step features (``/root/reference/rwgnn/rwgnn_pyg.py:94-153``), differentiable by autograd.

PARITY UNPINNED: the reference needs torch_geometric to run, which is not installed and not vendored, and it holds
no fixtures for this model.  The torch_geometric pieces (``RWConv`` = ``MessagePassing(aggr='add')`` with
``add_self_loops`` and ``degree``) are restated from the library's published semantics, as in
``oracle/rwconv_oracle.py``; everything else follows the reference's own operators line by line.  Without dropout.
"""
import torch


def rwconv(x, edge_index, weight, bias):
    """``rwgnn_pyg.py:25-46``: self-loops appended, message x[j] / deg[j] along j -> i, summed at i, then Linear."""
    n = x.shape[0]
    loops = torch.arange(n, dtype=edge_index.dtype)
    src = torch.cat([edge_index[0], loops])
    dst = torch.cat([edge_index[1], loops])
    deg = torch.bincount(src, minlength=n).to(x.dtype)
    deg_inv = deg.pow(-1.0)
    deg_inv[deg_inv == float("inf")] = 0
    msg = deg_inv[src].view(-1, 1) * x[src]
    agg = torch.zeros_like(x).index_add_(0, dst, msg)
    return agg @ weight.t() + bias


def rw_features(params, x, edge_index, batch, max_step, normalize=False):
    """params: adj_hidden (H, S(S-1)/2), features_hidden (H, S, D), fc.weight, fc.bias, and
    rw_layers.<i>.lin.weight / .bias.  Returns the (G, H*S*max_step) matrix of ``:153``."""
    adj_hidden, z = params["adj_hidden"], params["features_hidden"]
    h, s, _ = z.shape
    counts = torch.bincount(batch)
    g = counts.numel()
    ah = torch.zeros(h, s, s, dtype=z.dtype)
    iu = torch.triu_indices(s, s, 1)
    ah[:, iu[0], iu[1]] = torch.relu(adj_hidden)                                  # :105-110
    ah = ah + ah.transpose(1, 2)
    x = torch.sigmoid(x @ params["fc.weight"].t() + params["fc.bias"])            # :113
    zx = torch.einsum("abc,dc->abd", z, x)                                        # :115
    cur, out = x, []
    for i in range(max_step):
        if i > 0:
            cur = rwconv(cur, edge_index, params[f"rw_layers.{i - 1}.lin.weight"],
                         params[f"rw_layers.{i - 1}.lin.bias"])                   # :128
            z = torch.einsum("abc,acd->abd", ah, z)                               # :131
        t = zx * torch.einsum("abc,dc->abd", z, cur)                              # :125/:132, :135
        t_graphs = torch.zeros(h, s, g, dtype=t.dtype).index_add_(2, batch, t)    # :138-141
        t = t_graphs.transpose(0, 1)                                              # :144
        if normalize:
            t = t / counts.unsqueeze(1).repeat(1, h).unsqueeze(2)                 # :102, :146
        t = t.transpose(1, 2)                                                     # :148
        out.append(t.reshape(g, -1))                                              # :149
    return torch.cat(out, dim=1)
