import torch, sys
sys.path.insert(0, '/root/repo')
from celligner2_amd import ops
n = 4096*(1024+512+512+1024)
m = torch.empty(n, dtype=torch.uint8, device='cuda'); e = torch.empty(4096, 64, device='cuda')
st = torch.tensor([0,0], dtype=torch.int64, device='cuda')
for _ in range(5): ops.noise_dev(m, 0.05, e, 0, 1, st, advance=n//16)
torch.cuda.synchronize()
s=torch.cuda.Event(enable_timing=True); t=torch.cuda.Event(enable_timing=True)
s.record()
for _ in range(100): ops.noise_dev(m, 0.05, e, 0, 1, st, advance=n//16)
t.record(); torch.cuda.synchronize()
print("us per launch", s.elapsed_time(t)*10, st.tolist())
