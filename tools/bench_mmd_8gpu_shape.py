import sys, torch
sys.path.insert(0, "/root/repo")
from celligner2_amd import ops
Bg, Bl, D, C = 32768, 4096, 64, 3
torch.manual_seed(0)
z = torch.randn(Bg, D, device="cuda")
cond = torch.zeros(Bg, C, dtype=torch.int64, device="cuda"); cond[torch.arange(Bg), torch.randint(0, C, (Bg,))] = 1
part = torch.empty(Bg, dtype=torch.int32, device="cuda"); wt = torch.empty(64 * 64, device="cuda")
pws = torch.empty(int(ops.lib.c2_partition_ws(Bg)), dtype=torch.uint8, device="cuda"); st = torch.zeros(1, dtype=torch.int32, device="cuda")
def part_():
    ops.partition(cond, C, None, None, part, wt, pws, st)
loss = torch.zeros(1, device="cuda"); df = torch.empty(Bl, D, device="cuda"); ws = torch.empty(ops.mmd_ws(Bl, Bg, D), device="cuda")
def mmd_():
    ops.mmd(z, part, wt, 4096, Bl, 1.0, loss, df, ws)
for name, fn in (("partition", part_), ("mmd", mmd_)):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(10): fn()
    e.record(); torch.cuda.synchronize()
    print(f"{name} at Bg={Bg}, Bl={Bl}: {s.elapsed_time(e) / 10 * 1e3:.1f} us")
