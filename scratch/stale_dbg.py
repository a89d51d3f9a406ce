"""Who keeps an autograd graph alive into Trainer.capture()?  Mirrors tests/test_gpu_aux.py::
test_head_finish_folded_into_the_layer_backward and reports, for the loss of every eager step, whether its grad_fn
is still alive after the trainer dropped it, and what refers to it."""
import gc, os, sys, warnings, weakref
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import bench
from _golden import GoldenCase
import gnn_amd.function as fn
from gnn_amd import RW_NN
dev = torch.device("cuda:0")
c = GoldenCase("mutag_b0")
hp = dict(c.hp, dropout=0.0)

def trainer():
    torch.manual_seed(1)
    m = RW_NN(hp["input_dim"], hp["max_step"], hp["hidden_graphs"], hp["size_hidden_graphs"], hp["hidden_dim"],
              hp["penultimate_dim"], bool(hp["normalize"]), hp["n_classes"], hp["dropout"], dev).to(dev)
    m.load_state_dict(c.torch_state()); m.train(); m.seed_dropout(1)
    return bench.Trainer(m, c.torch_inputs(dev), 1)

gc.disable()
tr = trainer()
refs = []
for i in range(3):
    fn.FOLD_HEAD_FINISH = bool(int(sys.argv[1])) if len(sys.argv) > 1 else True
    tr.step_eager()
    refs.append(weakref.ref(tr.loss.grad_fn))
torch.cuda.synchronize()
print("alive after later steps replaced the loss:", [r() is not None for r in refs])
tr.loss = None
print("alive after loss=None:", [r() is not None for r in refs])
node = refs[-1]()
if node is not None:
    print("node:", node, "refcount", sys.getrefcount(node))
    for ref in gc.get_referrers(node):
        print("  referrer:", type(ref), (list(ref.keys()) if isinstance(ref, dict) else str(ref)[:200]))
        for ref2 in gc.get_referrers(ref):
            print("      <-", type(ref2), (list(ref2.keys())[:12] if isinstance(ref2, dict) else str(ref2)[:160]))
n = gc.collect()
print("gc.collect() found", n, "objects; alive after collect:", [r() is not None for r in refs])
with warnings.catch_warnings():
    warnings.simplefilter("error")
    try:
        tr.capture()
        print("capture: no warning")
    except Exception as exc:
        print("capture raised:", type(exc).__name__, str(exc)[:300])
