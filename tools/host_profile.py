"""cProfile of the launch path: N eager (not graph-replayed) training steps of the C4 bench model; prints the top functions by
own time.  usage: python tools/host_profile.py [steps]   (GPU box)"""
import cProfile
import os
import pstats
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from celligner2_amd.model import Celligner2  # noqa: E402
from celligner2_amd.trainer import Celligner2Trainer  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 60
cfg = dict(bench.CONFIGS["c4"])
cfg["N"] = 16384
dev = torch.device("cuda", 0)
torch.manual_seed(2020)
model = Celligner2(cfg["G"], cfg["hidden"], cfg["latent"], 0.05, False, "nb", conditions=[f"c{i}" for i in range(cfg["C"])],
                   use_mmd=True, mmd_on="z", beta=cfg["beta"], predictors=[f"p{i}" for i in range(cfg["P"])],
                   classifier_hidden_layer_sizes=cfg["cls_hidden"], precision="bf16")
data = bench.synth_dataset(cfg, dev)
comm = None
if os.environ.get("C2_DP_SINGLE_RANK") == "1":  # the data-parallel pipeline with one rank over RCCL
    import torch.distributed as dist
    for k, v in (("MASTER_ADDR", "127.0.0.1"), ("MASTER_PORT", "29871"), ("RANK", "0"), ("WORLD_SIZE", "1")):
        os.environ.setdefault(k, v)
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", device_id=dev)
    from celligner2_amd.parallel import DataParallelStep
    comm = DataParallelStep(shard_optimizer=True)
tr = Celligner2Trainer(model, None, train_data=data, valid_data=None, batch_size=cfg["B"], use_leiden=False, alpha_kl=1e-3,
                       monitor=False, use_early_stopping=False, rng="device", seed=2020, n_samples=16384, world=comm)
tr.initialize_loaders()
model.train()
tr.optimizer = tr.make_optimizer(lr=1e-3, eps=0.01)
tr.epoch, tr.iter = 0, 0
from collections import defaultdict  # noqa: E402
tr.iter_logs = defaultdict(list)


def batches():
    while True:
        for b in tr.dataloader_train:
            yield b


src = batches()
for _ in range(8):
    tr.on_iteration(next(src))
torch.cuda.synchronize()
import gc  # noqa: E402
gc.collect()
gc.freeze()
import time  # noqa: E402
t0 = time.perf_counter()
for _ in range(n):
    tr.on_iteration(next(src))
t1 = time.perf_counter()
torch.cuda.synchronize()
print(f"un-profiled: {1e3 * (t1 - t0) / n:.3f} ms host per step")
pr = cProfile.Profile()
pr.enable()
for _ in range(n):
    tr.on_iteration(next(src))
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(45)
st.sort_stats("cumtime").print_stats(40)
