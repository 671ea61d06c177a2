import sys, torch
sys.path.insert(0, "/root/repo")
from celligner2_amd import ops
B, D, C = 4096, 64, 3
torch.manual_seed(0)
for scale in (1.0, 0.05):
    z = torch.randn(B, D, device="cuda") * scale
    cond = torch.zeros(B, C, dtype=torch.int64, device="cuda"); cond[torch.arange(B), torch.randint(0, C, (B,))] = 1
    part = torch.empty(B, dtype=torch.int32, device="cuda"); wt = torch.empty(64 * 64, device="cuda")
    pws = torch.empty(int(ops.lib.c2_partition_ws(B)), dtype=torch.uint8, device="cuda"); st = torch.zeros(1, dtype=torch.int32, device="cuda")
    ops.partition(cond, C, None, None, part, wt, pws, st)
    loss = torch.zeros(1, device="cuda"); df = torch.empty(B, D, device="cuda"); ws = torch.empty(ops.mmd_ws(B, B, D), device="cuda")
    for _ in range(3): ops.mmd(z, part, wt, 0, B, 1.0, loss, df, ws)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(20): ops.mmd(z, part, wt, 0, B, 1.0, loss, df, ws)
    e.record(); torch.cuda.synchronize()
    print(f"scale {scale}: {s.elapsed_time(e) / 20 * 1e3:.1f} us per mmd call (incl. norms, sum_slabs, finalize)")
    if hasattr(ops.lib, "c2_dbg_mmd_clk"):
        import ctypes
        buf = (ctypes.c_ulonglong * 8)()
        ops.lib.c2_dbg_mmd_clk(buf)
        names = ["stage+barrier", "gram", "dist+min", "exps", "w/gw/ldsW", "contraction", "loop total"]
        print("   clk per phase (wave 0 of one block):", {n: int(buf[i]) for i, n in enumerate(names)})
