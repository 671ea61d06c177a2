import sys, os, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from celligner2_amd import ops
n = 19000 * 1024
ts = []
for _ in range(2):
    p, g, m, v = (torch.randn(n, device="cuda") for _ in range(4)); v.abs_()
    ts.append((p, g, m, v, torch.empty(n, dtype=torch.bfloat16, device="cuda")))
def run(): ops.adam_step(ts, 1e-3, 0.9, 0.999, 0.01, 0.04, 0.0, 3)
for _ in range(3): run()
torch.cuda.synchronize()
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record()
for _ in range(20): run()
e.record(); torch.cuda.synchronize()
us = s.elapsed_time(e) / 20 * 1e3
print(f"adam 2 x 19.5M params: {us:.1f} us  {2 * n * 30 / us / 1e6:.2f} TB/s")
