"""Micro-benchmark of the gene-width GEMMs at the C4 shapes (GPU box only)."""
import sys, os, time
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from celligner2_amd import ops

def bench(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / reps * 1e3  # us

def main():
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    which = sys.argv[2] if len(sys.argv) > 2 else "all"
    B, G, H = 4096, 19000, 1024
    ns = int(os.environ.get("NS", "2"))
    dev = "cuda"
    dt = torch.bfloat16
    x = torch.randn(B, G, device=dev).to(dt); w0 = torch.randn(H, G, device=dev).to(dt)
    d1 = torch.randn(B, H, device=dev).to(dt); wout = torch.randn(G, H, device=dev).to(dt)
    dl = torch.randn(B, G, device=dev).to(dt); dh0 = torch.randn(B, H, device=dev).to(dt)
    flops = 2.0 * B * G * H
    cases = {
        f"enc0_fwd  NT M=4096 N=1024 K=19000 ns={ns}": lambda y=torch.empty(ns, B, H, device=dev): ops.linear_fwd(x, w0, y, nslab=ns),
        "head_fwd  NT M=4096 N=19000 K=1024": lambda y=torch.empty(B, G, device=dev): ops.linear_fwd(d1, wout, y),
        f"head_dgrad NN M=4096 N(out)=1024 R=19000 ns={ns}": lambda y=torch.empty(ns, B, H, device=dev): ops.linear_dgrad(dl, wout, y, nslab=ns),
        "head_wgrad TN [19000,1024] R=4096": lambda y=torch.empty(G, H, device=dev): ops.linear_wgrad(dl, d1, y),
        "enc0_wgrad TN [1024,19000] R=4096": lambda y=torch.empty(H, G, device=dev): ops.linear_wgrad(dh0, x, y),
    }
    # the weight gradients with one (narrow) operand transposed: (XC, RC) and (RC, XC rows) on v_mfma_16x16x32
    d1t, dh0t = d1.t().contiguous(), dh0.t().contiguous()
    rows = torch.arange(B, device=dev, dtype=torch.int32)
    y_ref = torch.empty(G, H, device=dev)
    ops.linear_wgrad(dl, d1, y_ref)
    y_bt = torch.empty(G, H, device=dev)
    ops.linear_wgrad_bt(dl, d1t, y_bt)
    print("head_wgrad_bt vs wgrad: max rel diff", float((y_bt - y_ref).abs().max() / y_ref.abs().max()))
    y_ref2 = torch.empty(H, G, device=dev)
    ops.linear_wgrad_rows(dh0, x, rows, y_ref2)
    y_at = torch.empty(H, G, device=dev)
    ops.linear_wgrad_rows_at(dh0t, x, rows, y_at)
    print("enc0_wgrad_rows_at vs wgrad_rows: max rel diff", float((y_at - y_ref2).abs().max() / y_ref2.abs().max()))
    cases["head_wgrad_bt (XC,RC) [19000,1024] R=4096"] = lambda y=y_bt: ops.linear_wgrad_bt(dl, d1t, y)
    cases["enc0_wgrad_rows TN rows [1024,19000] R=4096"] = lambda y=y_ref2: ops.linear_wgrad_rows(dh0, x, rows, y)
    cases["enc0_wgrad_rows_at (RC,XC rows) [1024,19000]"] = lambda y=y_at: ops.linear_wgrad_rows_at(dh0t, x, rows, y)
    for name, fn in cases.items():
        if which != "all" and which not in name: continue
        us = bench(fn, reps)
        extra = ""
        if hasattr(ops.lib, "c2_dbg_clk"):  # clock-probe build (-DC2_GEMM_DBG=16): shader clock during the main loop
            import ctypes
            buf = (ctypes.c_ulonglong * 4)()
            ops.lib.c2_dbg_clk(buf)
            extra = f"  block0 loop: {buf[0]} clk in {buf[1] / 100:.1f} us = {buf[0] / max(buf[1], 1) * 100:.0f} MHz; wave0 wait {buf[2]} barrier {buf[3]} clk"
        print(f"{name:50s} {us:8.1f} us  {flops / us / 1e6:7.1f} TFLOP/s{extra}")

if __name__ == "__main__":
    main()
