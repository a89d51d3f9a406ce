"""gnn_amd -- MI355X-native random-walk-kernel layer (drop-in ``RW_NN``).

Only what the hot path needs: HIP kernels + C ABI (``csrc/``, ``librwk.so``), the graph-structure
cache, the autograd function, the ``nn.Module`` that mirrors the reference's interface, and the
graph-sharded data-parallel helpers, and the two steps that feed the path (``data``: TU-format loader and
device-side batch construction, ``utils.py:11-85``; ``rwconv`` / ``rw_gnn``: the row-normalised propagation
layer and the per-hidden-node model of ``rwgnn_pyg.py:19-170`` on the same kernels).
"""
from .model import RW_NN
from .structure import GraphStructure, clear_structure_cache, get_structure
from .function import rw_kernel_layer
from .optim import Adam

__all__ = ["RW_NN", "Adam", "GraphStructure", "get_structure", "clear_structure_cache", "rw_kernel_layer"]
