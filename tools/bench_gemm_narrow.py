"""Micro-benchmark of the narrow hidden-layer GEMMs (bf16 planes, 128x128 LDS-DMA kernel) at the C4 shapes."""
import sys, os
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
    return s.elapsed_time(e) / reps * 1e3

def main():
    dev = "cuda"
    B = 4096
    for (N, K, ns) in [(512, 1024, 2), (512, 1024, 1), (64, 512, 1), (64, 512, 4), (1024, 512, 1), (512, 64, 1)]:
        a = ops.split_bf16(torch.randn(B, K, device=dev)); w = ops.split_bf16(torch.randn(N, K, device=dev))
        dy = ops.split_bf16(torch.randn(B, N, device=dev))
        y = torch.empty(ns, B, N, device=dev)
        us = bench(lambda: ops.linear_fwd(a, w, y, nslab=ns))
        fl = 6.0 * B * N * K
        print(f"fwd   M={B} N={N} K={K} ns={ns}: {us:7.1f} us  {fl / us / 1e6:7.1f} TF(bf16-equiv)")
        da = torch.empty(ns, B, K, device=dev)
        us = bench(lambda: ops.linear_dgrad(dy, w, da, nslab=ns))
        print(f"dgrad M={B} N={N} K={K} ns={ns}: {us:7.1f} us  {fl / us / 1e6:7.1f} TF")
        nsw = 8
        dw = torch.empty(nsw, N, K, device=dev)
        us = bench(lambda: ops.linear_wgrad(dy, a, dw, nslab=nsw))
        print(f"wgrad M={B} N={N} K={K} ns={nsw}: {us:7.1f} us  {fl / us / 1e6:7.1f} TF")

if __name__ == "__main__":
    main()
