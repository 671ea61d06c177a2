"""Time c2_transpose_bf16 at the step's shape ([4096, 1024] bf16) and check it (GPU box)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from celligner2_amd import ops  # noqa: E402

for R, C in ((4096, 1024), (4096, 820), (1000, 520)):
    x = torch.randn(R, C, device="cuda").to(torch.bfloat16)
    y = torch.empty(C, R, device="cuda", dtype=torch.bfloat16)
    ops.transpose_bf16(x, y)
    assert torch.equal(y, x.t()), (R, C)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(50):
        ops.transpose_bf16(x, y)
    e.record()
    torch.cuda.synchronize()
    print(f"transpose {R}x{C}: {s.elapsed_time(e) * 20:.1f} us per call")
