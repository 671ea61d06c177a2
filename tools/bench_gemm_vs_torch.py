"""Reference point for the gene-width GEMMs: the same bf16 problems through torch.matmul (hipBLASLt / rocBLAS on this image),
fp32 output as the library's kernels produce it where stated.  GPU box only; not used by the product."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from celligner2_amd import ops  # noqa: E402

B, G, H = 4096, 19000, 1024
dev = "cuda"


def bench(fn, reps=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / reps * 1e3


x = torch.randn(B, G, device=dev).to(torch.bfloat16)
w0 = torch.randn(H, G, device=dev).to(torch.bfloat16)
d1 = torch.randn(B, H, device=dev).to(torch.bfloat16)
wout = torch.randn(G, H, device=dev).to(torch.bfloat16)
dl = torch.randn(B, G, device=dev).to(torch.bfloat16)
flops = 2.0 * B * G * H
y32 = torch.empty(B, G, device=dev)
cases = {
    "dec fwd   [4096,19000] = d1 @ wout^T   ours (fp32 out)": lambda: ops.linear_fwd(d1, wout, y32),
    "dec fwd   torch bf16 out": lambda: torch.matmul(d1, wout.t()),
    "dec fwd   torch fp32 out (out_dtype)": lambda: torch.mm(d1, wout.t(), out_dtype=torch.float32),
    "dec dgrad [4096,1024] = dl @ wout      ours (2 slabs)": lambda y=torch.empty(2, B, H, device=dev): ops.linear_dgrad(dl, wout, y, nslab=2),
    "dec dgrad torch bf16 out": lambda: torch.matmul(dl, wout),
    "dec wgrad [19000,1024] = dl^T @ d1     ours": lambda y=torch.empty(G, H, device=dev): ops.linear_wgrad(dl, d1, y),
    "dec wgrad ours, d1 transposed (bt)": lambda y=torch.empty(G, H, device=dev), t=d1.t().contiguous(): ops.linear_wgrad_bt(dl, t, y),
    "dec wgrad torch bf16 out": lambda: torch.matmul(dl.t(), d1),
    "enc fwd   [4096,1024] = x @ w0^T       ours (2 slabs)": lambda y=torch.empty(2, B, H, device=dev): ops.linear_fwd(x, w0, y, nslab=2),
    "enc fwd   torch bf16 out": lambda: torch.matmul(x, w0.t()),
    "enc wgrad [1024,19000] = dh^T @ x      ours": lambda y=torch.empty(H, G, device=dev): ops.linear_wgrad(d1, x, y),
    "enc wgrad torch bf16 out": lambda: torch.matmul(d1.t(), x),
}
for name, fn in cases.items():
    try:
        us = bench(fn)
        print(f"{name:58s} {us:8.1f} us  {flops / us / 1e6:7.1f} TFLOP/s")
    except Exception as ex:  # noqa: BLE001
        print(f"{name:58s} failed: {type(ex).__name__}: {str(ex)[:80]}")
