import sys, numpy as np, torch, torch.nn.functional as F
sys.path.insert(0,'.'); sys.path.insert(0,'tests')
from _golden import GoldenCase, rel_err
import test_gpu_parity as T
dev=torch.device('cuda:0')
c=GoldenCase('mutag_b2_norm'); m=T._model_for(c,dev)
adj,f,gi,y=c.torch_inputs(dev)
out=m(adj,f,gi); loss=F.cross_entropy(out,y); loss.backward()
truth=T._float64_port_grads(c)
for k in ['fc2.bias','fc2.weight','bn.bias','fc1.bias']:
    g=dict(m.named_parameters())[k].grad.cpu().numpy()
    print(k,'ours',g.ravel()[:4],'ref',c.grads[k].ravel()[:4],'truth',truth[k].ravel()[:4])
print('out err', rel_err(out.detach().cpu().numpy(), c['out']), 'loss', float(loss), float(c['loss']))
# same head on CPU from our K
k=m.rw_features(adj,f,gi).detach().cpu()
print('K ours vs ref', rel_err(k.numpy(), c['k']))
mc=T._model_for(c,'cpu')
kk=k.clone().requires_grad_(True)
o=mc.bn(kk); o=mc.relu(mc.fc1(o)); o=mc.fc2(o); o=F.log_softmax(o,1); l=F.cross_entropy(o,y.cpu()); l.backward()
print('cpu head on our K: fc2.bias', mc.fc2.bias.grad.numpy())
kk=torch.from_numpy(c['k']).clone().to(dev).requires_grad_(True)
m.zero_grad()
o=m.bn(kk); o=m.relu(m.fc1(o)); o=m.fc2(o); o=F.log_softmax(o,1); l=F.cross_entropy(o,y); l.backward()
print('gpu head on ref K: fc2.bias', m.fc2.bias.grad.cpu().numpy())
torch.backends.cudnn.enabled=False
kk=torch.from_numpy(c['k']).clone().to(dev).requires_grad_(True)
m.zero_grad()
o=m.bn(kk); o=m.relu(m.fc1(o)); o=m.fc2(o); o=F.log_softmax(o,1); l=F.cross_entropy(o,y); l.backward()
print('gpu head (miopen off) on ref K: fc2.bias', m.fc2.bias.grad.cpu().numpy(), 'out err', rel_err(o.detach().cpu().numpy(), c['out']))
torch.backends.cudnn.enabled=True
torch.backends.cuda.matmul.allow_tf32=False
for prec in ['highest']:
    torch.set_float32_matmul_precision(prec)
    kk=torch.from_numpy(c['k']).clone().to(dev).requires_grad_(True); m.zero_grad()
    o1=m.bn(kk); print('bn out err vs cpu', rel_err(o1.detach().cpu().numpy(), mc.bn(torch.from_numpy(c['k'])).detach().numpy()))
    o2=m.fc1(o1); print('fc1 err', rel_err(o2.detach().cpu().numpy(), mc.fc1(o1.detach().cpu()).detach().numpy()))
