"""Worker of tests/test_parallel_gpu.py::test_data_parallel_step_replayed_as_a_graph: a ONE-rank RCCL process group with
C2_DP_SINGLE_RANK=1 (the rank issues every collective of the data-parallel step on the one GPU of the box) trains the
same model twice -- the iteration launched kernel by kernel, and captured once and replayed as a HIP graph INCLUDING the
collectives (latent all-gather, bucket reduce-scatter / all-reduce on the side stream with the last bucket deferred into
the optimizer, parameter all-gather).  Same seeds, same rows, same draws: weights, optimizer step counts and epoch logs
must be the same bits.  argv[1]: "allreduce" | "shard" | "shard_bf16" | "bn"."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def run(variant, graph):
    from celligner2_amd.model import Celligner2
    from celligner2_amd.parallel import DataParallelStep
    from celligner2_amd.trainer import Celligner2Trainer
    from tests.test_trainer_gpu import make_adata
    bf16 = variant.endswith("bf16")
    G = 2304 if bf16 else 64
    np.random.seed(0)
    adata = make_adata(150, G, 3)
    torch.manual_seed(5)
    use_bn = variant == "bn"
    model = Celligner2(G, [32, 16], 8, 0.1, False, "nb", conditions=["ccle", "gtex", "tcga"], use_mmd=True, beta=3.0,
                       predictors=["lung", "skin", "blood"], use_bn=use_bn, use_ln=not use_bn,
                       precision="bf16" if bf16 else "fp32")
    comm = DataParallelStep(shard_optimizer=variant.startswith("shard"))
    assert comm.active and comm.sharded == variant.startswith("shard")
    tr = Celligner2Trainer(model, adata, condition_keys=["dataset"], predictor_keys=["lineage"], use_leiden=False,
                           batch_size=32, monitor=False, train_frac=0.8, seed=11, rng="device", use_early_stopping=False,
                           alpha_iter_anneal=7, graph=graph, world=comm)
    tr.train(n_epochs=4, lr=1e-2)
    return model, tr


def main():
    from tests.ddp_worker import _own_stderr
    _own_stderr()  # (this rank's error stream in a file of its own: an abort's message survives torchrun's tail)
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
    variant = sys.argv[1]
    m0, t0 = run(variant, False)
    m1, t1 = run(variant, True)
    ok = True
    if t1.graph_replays < 6 or t1.graph_refused is not None:
        ok = False
        print("the data-parallel step was not replayed as a graph:", t1.graph_replays, t1.graph_refused)
    for (k, a), (_, b) in zip(m0.state_dict().items(), m1.state_dict().items()):
        if not torch.equal(a, b):
            ok = False
            print(f"MISMATCH {variant}: {k} differs by {(a.float() - b.float()).abs().max().item():.3e}")
    for k in t0.logs:
        if not np.allclose(t0.logs[k], t1.logs[k], rtol=1e-6, atol=1e-9):
            ok = False
            print("MISMATCH logs", k, t0.logs[k], t1.logs[k])
    s0 = [t0.optimizer.state[p]["step"] for p in t0.optimizer.param_groups[0]["params"]]
    s1 = [t1.optimizer.state[p]["step"] for p in t1.optimizer.param_groups[0]["params"]]
    if s0 != s1:
        ok = False
        print("MISMATCH optimizer step counts", s0[:3], s1[:3])
    print("dp graph worker", variant, "ok" if ok else "FAILED", "replays", t1.graph_replays, flush=True)
    torch.cuda.synchronize()
    dist.barrier()
    dist.destroy_process_group()
    sys.exit(0 if ok else 1)


if __name__ == "__main__":
    main()
