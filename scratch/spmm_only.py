import sys, torch
sys.path.insert(0, '.')
import bench
from gnn_amd import get_structure
from gnn_amd.synth import CONFIGS, config_batch
dev = torch.device('cuda:0')
name, scale = sys.argv[1].split('@')
bt = bench.tile_batch(config_batch(name, 1).to_torch(dev), int(scale))
st = get_structure(bt[0], bt[2])
d = CONFIGS[name]['hp']['hidden_dim']
x = torch.rand((st.n, d), device=dev); y = torch.zeros((st.n, d), device=dev)
for _ in range(10):
    st.spmm(x, y)
torch.cuda.synchronize()
print(st.n, st.nnz, st.spmm_bytes(d))
