"""Loader for the committed golden vectors (made by oracle/make_golden.py from the reference)."""
import glob
import os

import numpy as np

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def case_names():
    return sorted(os.path.splitext(os.path.basename(p))[0] for p in glob.glob(os.path.join(GOLDEN_DIR, "*.npz")))


class GoldenCase:
    def __init__(self, name):
        self.name = name
        with np.load(os.path.join(GOLDEN_DIR, name + ".npz"), allow_pickle=False) as z:
            self.d = {k: z[k] for k in z.files}
        self.hp = {k[3:]: self.d[k].item() for k in self.d if k.startswith("hp.")}
        self.params = {k[6:]: self.d[k] for k in self.d if k.startswith("param.")}
        self.grads = {k[5:]: self.d[k] for k in self.d if k.startswith("grad.")}
        self.n_nodes = int(self.d["n_nodes"])
        self.train = bool(self.d["train"])

    def __getitem__(self, k):
        return self.d[k]

    def torch_inputs(self, device="cpu", dtype=None):
        import torch
        idx = torch.from_numpy(self.d["coo_indices"]).to(device)
        val = torch.from_numpy(self.d["coo_values"]).to(device)
        if dtype is not None:
            val = val.to(dtype)
        adj = torch.sparse_coo_tensor(idx, val, (self.n_nodes, self.n_nodes))
        feats = torch.from_numpy(self.d["features"]).to(device)
        if dtype is not None:
            feats = feats.to(dtype)
        gi = torch.from_numpy(self.d["graph_indicator"]).to(device)
        y = torch.from_numpy(self.d["y"]).to(device)
        return adj, feats, gi, y

    def torch_state(self):
        import torch
        return {k: torch.from_numpy(np.asarray(v)) for k, v in self.params.items()}


def rel_err(a, b):
    """max |a-b| / max(|b|): a norm-wise relative error (the scale of the tensor, not of each entry)."""
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    scale = np.max(np.abs(b)) if b.size else 0.0
    if scale == 0.0:
        return float(np.max(np.abs(a))) if a.size else 0.0
    return float(np.max(np.abs(a - b)) / scale)
