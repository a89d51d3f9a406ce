import sys; sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import torch, torch.nn.functional as F
from test_gpu_head import make_case
from gnn_amd.head import fused_head_loss, fused_head
from gnn_amd.losses import cross_entropy
dev = torch.device("cuda:0")
g, c, p, nc = 300, 32, 32, 3
m, m64, k, y = make_case(g, c, p, nc, True)
md = m.to(dev).eval(); yd = y.to(dev)
with torch.no_grad():
    out, loss = fused_head_loss(k.to(dev), yd, md.bn, md.fc1, md.fc2, 0.0, False)
    out2 = fused_head(k.to(dev), md.bn, md.fc1, md.fc2, 0.0, False)
    print(float(loss), float(cross_entropy(out, yd)), float(F.cross_entropy(out, yd)), float(F.cross_entropy(out.double(), yd)), torch.equal(out, out2))
    print(out[:4], yd[:4])
