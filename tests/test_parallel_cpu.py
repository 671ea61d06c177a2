"""World-size-2 gloo tests (CPU) of the data-parallel host logic: sharding, all-gather order, bucket all-reduce,
and the identity the design rests on -- per-rank losses/gradients normalised by the GLOBAL batch, with the MMD
gradient of a rank's own rows taken from (own rows x all rows), SUM to the single-process step.  The per-rank
compute in this test is the CPU oracle (test infrastructure); the product's kernels are covered on the GPU by
tests/test_parallel_gpu.py."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from celligner2_amd.parallel import DataParallelStep
    from oracle import c2_oracle as orc
    comm = DataParallelStep()
    assert comm.shard(10) == ((0, 5) if rank == 0 else (5, 10))
    assert comm.shard(11) == ((0, 5) if rank == 0 else (5, 11))
    # all-gather keeps rank order
    feat = torch.full((3, 2), float(rank))
    cond = torch.full((3, 2), rank, dtype=torch.int64)
    fa, ca = comm.allgather_rows(feat, cond)
    assert fa[:3].eq(0).all() and fa[3:].eq(1).all() and ca[3:].eq(1).all()
    # SyncBN: per-rank (mean, M2) partials gathered in rank order combine (Chan) to the statistics of all rows
    rows = np.random.default_rng(5).standard_normal((10, 4)) * 3 + 1
    lo_, hi_ = comm.shard(10)
    mine_ = rows[lo_:hi_]
    part = torch.tensor(np.stack([mine_.mean(0), ((mine_ - mine_.mean(0)) ** 2).sum(0)]))
    allp = comm.allgather_stats(part).numpy()
    assert allp.shape == (2, 2, 4) and np.allclose(allp[rank], part.numpy())
    n0, n1 = 5, 5
    delta = allp[1, 0] - allp[0, 0]
    mean = allp[0, 0] + delta * n1 / (n0 + n1)
    m2 = allp[0, 1] + allp[1, 1] + delta ** 2 * n0 * n1 / (n0 + n1)
    assert np.allclose(mean, rows.mean(0)) and np.allclose(m2 / 10, rows.var(0))
    # bucket all-reduce = SUM
    g = torch.arange(6, dtype=torch.float32) * (rank + 1)
    comm.allreduce_bucket_async(g)
    comm.wait_all()
    assert torch.equal(g, torch.arange(6, dtype=torch.float32) * 3)
    # sharded form: bucket chunks, the Adam entries cut down to the owned elements, the parameter all-gather
    sh = DataParallelStep(shard_optimizer=True)
    assert sh.sharded
    gflat = torch.zeros(64 * world * 3)
    pflat = torch.arange(64 * world * 3, dtype=torch.float32) + 1000 * rank  # every rank "updated" its own chunk
    buckets = [(0, 64 * world), (64 * world, 64 * world * 3)]
    sh.set_arena(gflat, pflat, buckets)
    own = sh.owned_ranges()
    base = gflat.data_ptr()
    assert own == [(base + 4 * 64 * rank, base + 4 * 64 * (rank + 1)),
                   (base + 4 * (64 * world + 128 * rank), base + 4 * (64 * world + 128 * (rank + 1)))]
    # a "parameter" whose gradient spans the boundary between the two ranks' chunks of bucket 0 and runs into bucket 1
    g = gflat[40:64 * world + 30]
    pp, mm, vv = torch.zeros(g.numel()), torch.zeros(g.numel()), torch.zeros(g.numel())
    parts = sh.owned_parts([(pp, g, mm, vv, None)])
    got = sorted((q[1].data_ptr() - base) // 4 for q in parts), sorted(q[1].numel() for q in parts)
    want = {0: ([40, 128], [24, 30]), 1: ([64], [64])}[rank]  # rank 0: tail of its chunk + head of its bucket-1 chunk
    assert got == (sorted(want[0]), sorted(want[1])), (rank, got)
    assert all(q[0].data_ptr() - pp.data_ptr() == q[1].data_ptr() - g.data_ptr() for q in parts)
    sh.allgather_params()
    for lo, hi in buckets:
        n = (hi - lo) // world
        for r in range(world):
            seg = pflat[lo + r * n:lo + (r + 1) * n]
            assert torch.equal(seg, torch.arange(lo + r * n, lo + (r + 1) * n, dtype=torch.float32) + 1000 * r)
    # the sharded optimizer's state_dict(): a collective that returns the complete moments on every rank -- each rank's
    # own elements from that rank, whatever stale values the others hold there
    from celligner2_amd.optim import FusedAdam
    pa, pb = pflat[40:64 * world + 30], pflat[64 * world + 30:64 * world * 3]  # two "parameters", views of the flat buffer
    opt = FusedAdam([torch.nn.Parameter(pa), torch.nn.Parameter(pb)], grad_sync=sh)
    plist = [q for g_ in opt.param_groups for q in g_["params"]]
    assert [sh.owned_slices(q) for q in plist] == {0: [[(0, 24), (88, 118)], [(0, 98)]], 1: [[(24, 88)], [(98, 226)]]}[rank]
    for q in plist:  # every rank's moments: 100 * (rank + 1) + element index inside its own elements, junk elsewhere
        mine_m = torch.full((q.numel(),), -7.0)
        for i0, i1 in sh.owned_slices(q):
            mine_m[i0:i1] = 100.0 * (rank + 1) + torch.arange(i0, i1, dtype=torch.float32)
        opt.state[q] = dict(step=5, exp_avg=mine_m, exp_avg_sq=2 * mine_m)
    sd = opt.state_dict()
    for i, want_owner in enumerate(([0] * 24 + [1] * 64 + [0] * 30, [0] * 98 + [1] * 128)):
        want = torch.tensor([100.0 * (o + 1) + k for k, o in enumerate(want_owner)])
        assert sd["state"][i]["step"] == 5
        assert torch.equal(sd["state"][i]["exp_avg"], want) and torch.equal(sd["state"][i]["exp_avg_sq"], 2 * want), (rank, i)
        assert torch.equal(opt.state[plist[i]]["exp_avg"][:1], torch.tensor([-7.0 if rank == 1 else 100.0])), "state untouched"
    # the decomposition identity, with the oracle as the per-rank compute
    rng = np.random.default_rng(0)
    B, G, C, P, L = 16, 40, 2, 3, 4
    cfg = orc.Cfg(input_dim=G, hidden_layer_sizes=[12, 8], latent_dim=L, n_conditions=C, n_predictors=P,
                  recon_loss="nb", use_mmd=False)
    params = orc.init_params(cfg, 0, np.float64)
    x = rng.poisson(3.0, (B, G)).astype(np.float64)
    condm = np.zeros((B, C), np.int64)
    condm[np.arange(B), rng.integers(0, C, B)] = 1
    cls = np.zeros((B, P), np.int64)
    cls[np.arange(B), rng.integers(0, P, B)] = 1
    eps = rng.standard_normal((B, L))
    full = dict(x=x, batch=condm, sizefactor=x.sum(1), goodloc=np.ones((B, G), bool), classes=cls)
    ref = orc.forward_backward(params, cfg, full, eps, gscale=(1, 1e-3, 1, 1))
    lo, hi = comm.shard(B)
    mine = {k: v[lo:hi] for k, v in full.items()}
    loc = orc.forward_backward(params, cfg, mine, eps[lo:hi], gscale=(1, 1e-3, 1, 1))
    # local recon / KL are means over the LOCAL rows: rescale to the global batch as the kernels do (inv_B)
    share = (hi - lo) / B
    scale = {"recon": share, "kl": share}
    losses = torch.tensor([loc["losses"][0] * share, loc["losses"][1] * share, loc["losses"][2], loc["losses"][3]])
    comm.allreduce_scalars(losses)
    assert np.allclose(losses.numpy(), ref["losses"], rtol=1e-10)
    # gradients: the CE part is a sum (no rescale); recon/KL parts scale with `share`.  Check on theta (pure recon).
    gth = torch.tensor(loc["grads"]["theta"] * share)
    comm.allreduce_bucket_async(gth)
    assert np.allclose(gth.numpy(), ref["grads"]["theta"], rtol=1e-9, atol=1e-12)
    # MMD: the gradient of the global MMD w.r.t. a rank's rows is the slice of the global gradient
    z = rng.standard_normal((B, L))
    m_all, d_all = orc.mmd_loss(z, condm, C, 2.0, None, None)
    out[rank] = float(np.abs(d_all[lo:hi]).sum())
    dist.destroy_process_group()


def test_data_parallel_host_logic_gloo_ws2():
    port = 29500 + (os.getpid() % 1000)
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(2, port, out), nprocs=2, join=True)
    assert len(out) == 2


def test_loader_shards_global_batches():
    from celligner2_amd.trainer import _Loader

    class FakeDS:
        device = torch.device("cpu")

        def __len__(self):
            return 23

        def gather(self, idx, *a, **kw):
            return idx

    class W:
        def __init__(self, r):
            self.rank, self.world_size = r, 2

    torch.manual_seed(3)
    a = [b.tolist() for b in _Loader(FakeDS(), 8, None, None, "reference", world=W(0))]
    torch.manual_seed(3)
    b = [b.tolist() for b in _Loader(FakeDS(), 8, None, None, "reference", world=W(1))]
    torch.manual_seed(3)
    full = [c.tolist() for c in _Loader(FakeDS(), 8, None, None, "reference")]
    assert len(a) == len(b) == len(full) == 3
    for fa, fb, ff in zip(a, b, full):
        per = len(ff) // 2
        assert fa == ff[:per] and fb == ff[per:2 * per]


def test_sharded_optimizer_state_before_the_first_step_is_whole():
    """shard_optimizer=True before any backward pass built the arena: nothing is partial yet, state_dict() is the plain one
    (the consolidation over ranks is covered by the two-rank worker above and by tests/ddp_worker.py on the GPU)."""
    import torch
    from celligner2_amd.optim import FusedAdam

    class Comm:
        defer_last = False

        def __init__(self, sharded):
            self.sharded = sharded

        def owned_slices(self, p):
            return None

    p = torch.nn.Parameter(torch.zeros(4))
    opt = FusedAdam([p], grad_sync=Comm(True))
    opt.state[p] = dict(step=3, exp_avg=torch.ones(4), exp_avg_sq=torch.full((4,), 2.0))
    sd = opt.state_dict()
    assert sd["state"][0]["step"] == 3 and torch.equal(sd["state"][0]["exp_avg"], torch.ones(4))
    sd = FusedAdam([p], grad_sync=Comm(False)).state_dict()
    assert sd["param_groups"][0]["lr"] == 1e-3 and FusedAdam([p]).state_dict()["state"] == {}
