import sys; sys.path.insert(0,'.'); sys.path.insert(0,'tests')
import torch, numpy as np
from _golden import GoldenCase
import importlib.util
spec=importlib.util.spec_from_file_location('ta','tests/test_gpu_aux.py'); ta=importlib.util.module_from_spec(spec); spec.loader.exec_module(ta)
dev=torch.device('cuda:0')
c=GoldenCase('mutag_b0'); hp=dict(c.hp,dropout=0.0)
mode=sys.argv[1]
g=ta._trainer(c,hp,dev); e=ta._trainer(c,hp,dev)
if mode=='unfused': g.model.fused_head=False; e.model.fused_head=False
if mode=='dropref':
    # warm up on the side stream, drop the old autograd graph, then capture
    s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(3): g.step_eager()
    torch.cuda.current_stream().wait_stream(s); torch.cuda.synchronize()
    g.loss=None
    g.g_fb=torch.cuda.CUDAGraph()
    with torch.cuda.graph(g.g_fb): g.fwd_bwd()
    g.g_opt=torch.cuda.CUDAGraph()
    with torch.cuda.graph(g.g_opt): g.opt.step()
else:
    g.capture()
for _ in range(3): e.step_eager()
for s in range(3):
    e.fwd_bwd(); g.g_fb.replay(); torch.cuda.synchronize()
    bad=[(n,float((pa.grad-pb.grad).abs().max())) for (n,pa),(_,pb) in zip(e.model.named_parameters(),g.model.named_parameters()) if float((pa.grad-pb.grad).abs().max())>0]
    print(mode,'step',s,'differing grads',bad)
    e.opt.step(); g.g_opt.replay()
