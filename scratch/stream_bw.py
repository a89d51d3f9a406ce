"""What a plain streaming kernel reaches on this device: elementwise copy / add of 1 GiB operands, and a read-only sum."""
import torch
dev = torch.device("cuda:0")
n = 1 << 28
a = torch.rand(n, device=dev); b = torch.empty_like(a)
def timeit(fn, it=20):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / it * 1e-3
t = timeit(lambda: b.copy_(a)); print("copy 1 GiB -> 1 GiB: %.1f us, %.2f TB/s" % (t * 1e6, 2 * n * 4 / t / 1e12))
t = timeit(lambda: torch.add(a, 1.0, out=b)); print("add: %.2f TB/s" % (2 * n * 4 / t / 1e12))
t = timeit(lambda: a.sum()); print("sum (read only): %.2f TB/s" % (n * 4 / t / 1e12))
t = timeit(lambda: b.fill_(1.0)); print("fill (write only): %.2f TB/s" % (n * 4 / t / 1e12))
