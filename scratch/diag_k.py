import sys, numpy as np, torch
sys.path.insert(0,'.'); sys.path.insert(0,'tests')
from _golden import GoldenCase, case_names, rel_err
from oracle import rwnn_oracle as O
import test_gpu_parity as T
dev=torch.device('cuda:0')
for n in case_names():
    c=GoldenCase(n); m=T._model_for(c,dev)
    adj,f,gi,_=c.torch_inputs(dev)
    k=m.rw_features(adj,f,gi).detach().cpu().numpy()
    csr=(c['csr_crow'],c['csr_col'],c['csr_val'])
    kr=O.rw_kernel_forward(csr,T._x0(c),c.params['adj_hidden'],c.params['features_hidden'],c['graph_indicator'],c.hp['max_step'],c.hp['normalize'])
    # elementwise relative error
    er_ours=np.max(np.abs(k-kr)/np.abs(kr).clip(1e-30)); er_ref=np.max(np.abs(c['k']-kr)/np.abs(kr).clip(1e-30))
    print(n,'ours norm %.2e elem %.2e | ref norm %.2e elem %.2e'%(rel_err(k,kr),er_ours,rel_err(c['k'],kr),er_ref))
