"""Odd model shapes through the data-parallel step on ONE GPU (GPU box; test infrastructure): tests/ddp_worker.py under
torch.distributed.run with 2, 4 and 8 gloo ranks sharing the card, every variant of the exchange (all-reduce, sharded
optimizer in fp32 and bf16 mode, SyncBN), each compared by the worker with the single-process step on the global batch.
The shapes exercise the gradient arena's bucket padding and the cut of every bucket into `world` aligned chunks.

    python tools/fuzz_ddp.py --cases 6 --seed 0

Seen once in ~100 launches (8 ranks on the one card, shape 2193,28,72,122,111,94): one rank ended with SIGABRT after
rank 0 had already printed its passing comparison, i.e. while the processes were shutting down; the same launch passed
when run again, and nothing in its output named a cause.  Recorded here, not explained.
"""
import argparse
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=6)
    ap.add_argument("--seed", type=int, default=0)
    a = ap.parse_args()
    rng = np.random.default_rng(a.seed)
    fails = 0
    port = 29700
    for i in range(a.cases):
        world = int(rng.choice([2, 4, 8]))
        variant = ["ln", "bn", "shard", "shard_bf16"][i % 4]
        G = int(rng.integers(2100, 2600)) if variant.endswith("bf16") else int(rng.integers(100, 700))
        hidden = [int(rng.integers(9, 130)) for _ in range(int(rng.integers(2, 4)))]
        L = int(rng.integers(8, 40))
        B = world * int(rng.integers(8, 40))  # equal shares (the MMD all-gather needs that)
        shape = ",".join(str(v) for v in [G, L, B] + hidden)
        env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", C2_DDP_SHAPE=shape, C2_DDP_PARAM_TOL_LR="0.15",
                   # two steps: the second one runs on the exchanged / all-gathered parameters.  More steps at these small
                   # per-rank batches compare two trajectories, not two implementations: one unit on the other side of a
                   # ReLU (kernels 1e-6 apart) is 1/24 of a gradient at 24 rows per rank, and the runs drift apart from
                   # there (traced with C2_DDP_DUMP=1 on shape 2459,23,48,128,62: equal to 2e-5 after two steps, first
                   # sign disagreement at step 3).  The pinned tests keep six steps at their fixed shape.
                   C2_DDP_STEPS="2", C2_DDP_LOG_DIR=os.path.join(ROOT, "gpurun_out", "ddp_logs", f"fuzz{i}"))
        port += 1
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr",
               "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "tests", "ddp_worker.py"), variant]
        t0 = time.time()
        r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=170)
        ok = r.returncode == 0
        fails += not ok
        print(f"[{i}] world {world} {variant} G,L,B,hidden = {shape}: {'ok' if ok else 'FAIL'} ({time.time() - t0:.0f}s)", flush=True)
        if not ok:
            print((r.stdout[-2500:] + r.stderr[-2500:]), flush=True)
            logs = env["C2_DDP_LOG_DIR"]  # every rank's own error stream (an abort's message survives here)
            for f in sorted(os.listdir(logs)) if os.path.isdir(logs) else []:
                print(f"--- {f}\n" + open(os.path.join(logs, f)).read()[-2500:], flush=True)
    print(f"{fails} failure(s)")
    sys.exit(min(fails, 100))


if __name__ == "__main__":
    main()
