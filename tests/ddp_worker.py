"""Worker of tests/test_parallel_gpu.py: 2 ranks share cuda:0 (gloo backend, the box has one GPU); each runs the
data-parallel step on its half of a global batch; rank 0 also runs the single-process step on the whole batch.
The summed-gradient result must equal the single-process one (fp32 mode, 1e-4).  argv[1] == "bn": the BatchNorm1d
variant, whose statistics are exchanged between the ranks (SyncBN); the running statistics must match too.
argv[1] == "shard" / "shard_bf16": the gradient buckets are reduce-scattered, each rank runs Adam on its own chunks and the
parameters are all-gathered (DataParallelStep(shard_optimizer=True)); the parameters after 6 steps must equal the
single-process run's."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _own_stderr():
    """Every rank keeps its own error stream: torchrun shows the tail of ONE merged stream, so the message of a rank that
    aborts (a C++ `terminate`, a HIP runtime error) was lost once.  C2_DDP_LOG_DIR=<dir>: file
    descriptor 2 of this rank goes to <dir>/rank<r>.err; the launcher prints those files when the run fails.  The Python
    stacks of all threads are dumped there on SIGABRT / SIGSEGV."""
    import faulthandler
    d = os.environ.get("C2_DDP_LOG_DIR")
    if d:
        os.makedirs(d, exist_ok=True)
        fd = os.open(os.path.join(d, f"rank{os.environ.get('RANK', '0')}.err"), os.O_WRONLY | os.O_CREAT | os.O_TRUNC, 0o644)
        sys.stderr.flush()
        os.dup2(fd, 2)
        os.close(fd)
    faulthandler.enable(all_threads=True)


def _finish(ok, on_device):
    """Orderly end of a rank: agree on the verdict, drain this rank's GPU queues (side stream included), meet the peers
    at a barrier so that no rank closes its sockets while another is still inside a collective, only then destroy the
    process group and leave.  (Without the barrier a fast rank's `destroy_process_group` + exit closes gloo pairs that a
    slower peer is still reading from: the peer's transport thread raises in a context nobody catches -> SIGABRT.)"""
    flag = torch.tensor([1.0 if ok else 0.0], device="cuda" if on_device else "cpu")
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    verdict = flag.item() == 1.0
    torch.cuda.synchronize()
    dist.barrier()
    dist.destroy_process_group()
    sys.stdout.flush()
    sys.stderr.flush()
    sys.exit(0 if verdict else 1)


def main():
    _own_stderr()
    # C2_DDP_BACKEND=nccl with ONE rank (and C2_DP_SINGLE_RANK=1, so that the rank still issues every collective): the RCCL
    # branch of parallel.py -- stream-ordered exchange on the side stream, deferred last bucket, reduce-scatter and
    # all-gather through the staging chunk -- on the one GPU there is.
    backend = os.environ.get("C2_DDP_BACKEND", "gloo")
    torch.cuda.set_device(0)
    if backend == "nccl":
        dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
    else:
        dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    from celligner2_amd.model import Celligner2
    from celligner2_amd.parallel import DataParallelStep
    from oracle import c2_oracle as orc

    if len(sys.argv) > 1 and sys.argv[1] == "noise":
        return noise_variant(rank, world)
    variant = sys.argv[1] if len(sys.argv) > 1 else ""
    use_bn = variant == "bn"
    shard = variant.startswith("shard")
    precision = "bf16" if variant.endswith("bf16") else "fp32"
    global NSTEPS
    NSTEPS = int(os.environ.get("C2_DDP_STEPS", "6"))
    GSTEP = int(os.environ.get("C2_DDP_GSTEP", "0"))  # (debugging: the step whose losses / gradients are compared)
    G, hidden, L, C, P, B = (2304 if precision == "bf16" else 256), [48, 24], 8, 3, 4, 64
    if os.environ.get("C2_DDP_SHAPE"):  # tools/fuzz_ddp.py: "G,L,B,h1,h2[,h3]" -- odd widths through the bucket / chunk bookkeeping
        v = [int(t) for t in os.environ["C2_DDP_SHAPE"].split(",")]
        G, L, B, hidden = v[0], v[1], v[2], v[3:]
    cfg = orc.Cfg(input_dim=G, hidden_layer_sizes=hidden, latent_dim=L, n_conditions=C, n_predictors=P,
                  recon_loss="nb", use_mmd=True, beta=20.0, use_bn=use_bn)
    rng = np.random.default_rng(0)
    params = orc.init_params(cfg, 0, np.float32)
    x = rng.poisson(rng.uniform(0.5, 2, (B, 1)) * np.exp(rng.normal(1, 1, (1, G)))).astype(np.float32)
    cond = np.zeros((B, C), np.int64)
    cond[np.arange(B), rng.integers(0, C, B)] = 1
    cls = np.zeros((B, P), np.int64)
    cls[np.arange(B), rng.integers(0, P, B)] = 1
    cls[::7] = -1
    full = dict(x=x, batch=cond, sizefactor=x.sum(1), goodloc=rng.uniform(size=(B, G)) > 0.01, classes=cls)
    eps = rng.standard_normal((B, L)).astype(np.float32)

    def build():
        m = Celligner2(G, hidden, L, 0.0, False, "nb", use_bn, conditions=["a", "b", "c"], use_mmd=True, beta=20.0,
                       predictors=list("wxyz"), precision=precision)
        m.load_state_dict({k: torch.tensor(v) for k, v in params.items()}, strict=False)  # BN buffers: defaults
        return m.cuda().train()

    from celligner2_amd.optim import FusedAdam

    def run(m, sl, sync=None):
        """One full step (forward, backward, fused Adam).  sync: the communicator -- the optimizer then updates the
        tensors of the bucket that is still being all-reduced in a second launch, after the wait."""
        opt = FusedAdam(list(m.parameters()), lr=1e-3, eps=0.01, weight_decay=0.04, grad_sync=sync)
        m._forced_noise = dict(eps=torch.tensor(eps[sl]).cuda(), masks={})
        first = None
        for step in range(NSTEPS):  # several steps back to back, nothing synchronised in between
            opt.zero_grad()
            if os.environ.get("C2_DDP_FORCE_REFRESH") in ("both", "ddp" if sync is not None else "ref") and step:
                eng = m._engine()  # (debugging: drop every operand copy of the weights before the forward)
                eng._shadow_ver = {k_: None for k_ in eng._shadow_ver}
                eng._wplanes.clear()
            r, k, mm, ce = m(**{k_: torch.tensor(v[sl]).cuda() for k_, v in full.items()})
            (r + 1e-3 * k + mm + ce).backward()
            if step == GSTEP:  # losses / gradients of the first step are compared; parameters after the last
                first = (torch.stack([r, k, mm, ce]).detach().clone(),
                         {n: p.grad.clone() for n, p in m.named_parameters() if p.grad is not None})
            opt.step()
        torch.cuda.synchronize()
        return first[0].cpu().numpy().astype(np.float64), first[1], opt

    ref_l = ref_g = ref_m = None
    if rank == 0:
        ref_m = build()
        ref_l, ref_g, ref_opt = run(ref_m, slice(0, B))
    m = build()
    comm = DataParallelStep(shard_optimizer=shard)
    comm.attach_model(m)
    per = B // world
    losses, grads, opt = run(m, slice(rank * per, (rank + 1) * per), sync=comm)
    assert comm.defer_last and comm.deferred_range() is None
    assert comm._host_backend == (backend != "nccl") and m._engine().exchange
    if shard:
        eng = m._engine()
        assert comm.sharded and eng.param_flat is not None
        lo_, hi_ = eng.param_flat.data_ptr(), eng.param_flat.data_ptr() + 4 * eng.param_flat.numel()
        assert all(lo_ <= p.data_ptr() < hi_ for p in m.parameters()), "parameters are views of the flat buffer"
        assert all((hi - lo) % (64 * world) == 0 for lo, hi in eng.buckets)
    m._engine().check_status()
    sd = opt.state_dict()  # (sharded form: a collective -- every rank gets the complete moments)
    if os.environ.get("C2_DDP_DUMP") and rank == 0:  # (debugging: the workspaces of the last backward, rank 0's rows)
        we, wr = m._engine()._ws, ref_m._engine()._ws
        for key in sorted(we, key=str):
            a, b = we[key], wr.get(key)
            if not (torch.is_tensor(a) and torch.is_tensor(b) and a.is_floating_point()):
                continue
            if a.dim() == 3 and b.dim() == 3 and a.shape[1] == per and b.shape[1] == B:
                a, b = a.float().sum(0), b.float().sum(0)[:per]
            elif a.dim() == 2 and b.dim() == 2 and a.shape[0] == per and b.shape[0] == B and a.shape[1] == b.shape[1]:
                a, b = a.float(), b.float()[:per]
            else:
                continue
            e = float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))
            flips = int(((a > 0) != (b > 0)).sum()) if str(key).endswith("yhat") else -1
            print(f"   ws {key}: {tuple(a.shape)} {e:.3g}" + (f"  sign disagreements {flips}" if flips >= 0 else ""))
    ok = True
    if rank == 0:
        tol_g, tol_p = (2e-2, 1e-4) if precision == "bf16" else (1e-4, 1e-5)  # bf16: operand copies may round apart
        ok = bool(np.allclose(losses, ref_l, rtol=1e-4 if precision == "fp32" else 1e-3, atol=1e-5))
        for n, b in ref_m.named_buffers():  # SyncBN: every rank holds the running statistics of the global batch
            err = (dict(m.named_buffers())[n].double() - b.double()).abs().max().item() / max(b.abs().max().item(), 1e-30)
            if err > 1e-5:
                ok = False
                print("MISMATCH buffer", n, err)
        for n in ref_g:
            den = ref_g[n].abs().max().item()
            if use_bn and n.endswith(".bias") and n[:-4] + "weight" in ref_g:  # cancelling batch sums (gu.grad_err)
                den = max(den, ref_g[n[:-4] + "weight"].abs().max().item())
            err = (grads[n] - ref_g[n]).abs().max().item() / max(den, 1e-30)
            worst_g = max(locals().get("worst_g", 0.0), err)
            if os.environ.get("C2_DDP_VERBOSE"):
                print(f"   grad {n}: {err:.3g}")
            if err > tol_g:
                ok = False
                print("MISMATCH", n, err)
        for n, p in ref_m.named_parameters():  # parameters after the optimizer step
            dabs = (dict(m.named_parameters())[n].detach() - p.detach()).abs().max().item()
            err = dabs / max(p.abs().max().item(), 1e-30)
            # (second bound, in units of the step: Adam divides every entry by its own |g| + eps, so an entry with
            # |g| << eps = 0.01 turns a summation-order difference d of the gradient into lr * d / eps -- with gradients
            # up to ~100 and d ~ 2.5e-7 of that, 0.25 % of lr per step, seen at 8 ranks x 2502 genes.  A chunk that is
            # skipped or updated twice moves by ~lr per step, 50x this bound.)
            # (tools/fuzz_ddp.py widens it to 0.15 lr per step: one unit of one row on the other side of a ReLU than in
            # the single-process run -- their kernels differ by ~1e-6 in bf16x3 mode, counted with C2_DDP_DUMP=1 -- changes
            # that row's gradient, ~1/B of every tensor upstream, and has been seen to move small-|g| entries by 0.06 lr)
            if err > tol_p and dabs > float(os.environ.get("C2_DDP_PARAM_TOL_LR", "2e-2")) * 1e-3 * NSTEPS:
                ok = False
                print("MISMATCH param after Adam", n, err)
        # Adam's moments after the last step: whole on this rank (in the sharded form through state_dict()'s SUM over ranks)
        ref_sd = ref_opt.state_dict()
        assert sorted(sd["state"]) == sorted(ref_sd["state"])
        for i, st in ref_sd["state"].items():
            for key in ("exp_avg", "exp_avg_sq"):
                den = st[key].abs().max().item()
                err = (sd["state"][i][key] - st[key]).abs().max().item() / max(den, 1e-30)
                if sd["state"][i]["step"] != st["step"] or err > (5e-2 if precision == "bf16" else 1e-3):
                    ok = False
                    print("MISMATCH optimizer state", i, key, err, sd["state"][i]["step"], st["step"])
        print("losses ddp", losses, "single", ref_l, "worst first-step gradient error", locals().get("worst_g"))
    _finish(ok, backend == "nccl")


def noise_variant(rank, world):
    """eps and dropout masks of a data-parallel step: different on every rank (rows of ONE global draw), while the
    shared generators stay in lock-step (the index stream must be the same on all ranks)."""
    from celligner2_amd.model import Celligner2
    from celligner2_amd.parallel import DataParallelStep

    B, L = 32, 8
    ok = True
    for mode in ("reference", "device"):
        torch.manual_seed(77)
        torch.cuda.manual_seed(77)
        m = Celligner2(64, [48, 24], L, 0.2, False, "nb", conditions=["a", "b"], predictors=["p", "q"]).cuda().train()
        m.rng_mode = mode
        comm = DataParallelStep()
        comm.attach_model(m)
        torch.manual_seed(5)
        torch.cuda.manual_seed(5)
        eps, masks = m._draw_noise(B, True, torch.device("cuda", 0))
        eps, masks = eps.clone(), {k: v.clone() for k, v in masks.items()}  # (the device masks are views of one buffer)
        eps2, masks2 = m._draw_noise(B, True, torch.device("cuda", 0))  # a second step: fresh draws
        after = torch.randperm(50).tolist() if mode == "reference" else torch.randperm(50, device="cuda").tolist()
        mine = dict(eps=eps.cpu().numpy(), eps2=eps2.cpu().numpy(), after=after,
                    masks={k: v.cpu().numpy() for k, v in masks.items()},
                    masks2={k: v.cpu().numpy() for k, v in masks2.items()})
        got = [None] * world
        dist.all_gather_object(got, mine)
        if rank == 0:
            a, b = got[0], got[1]
            bad = []

            def chk(what, cond):
                if not bool(cond):
                    bad.append(what)

            assert sorted(a["masks"]) == ["cls.D0", "dec.D0", "dec.D1", "enc.D0", "enc.D1"], sorted(a["masks"])
            chk("eps differs between ranks", a["eps"].shape == (B, L) and not np.allclose(a["eps"], b["eps"]))
            chk("eps differs between steps", not np.allclose(a["eps"], a["eps2"]))
            for k in a["masks"]:
                chk(k + " differs between ranks",
                    a["masks"][k].shape == b["masks"][k].shape and not np.array_equal(a["masks"][k], b["masks"][k]))
                chk(k + " differs between steps", not np.array_equal(a["masks"][k], a["masks2"][k]))
                chk(k + " rank 0 step 2 differs from rank 1 step 1", not np.array_equal(a["masks2"][k], b["masks"][k]))
                chk(k + f" keep rate {a['masks'][k].mean():.3f}", 0.7 < a["masks"][k].mean() < 0.9)
            chk("generators in lock-step after the draws", a["after"] == b["after"])
            if mode == "reference":  # the ranks hold the rows of the single-process draw on the global batch
                torch.manual_seed(5)
                m1 = Celligner2(64, [48, 24], L, 0.2, False, "nb", conditions=["a", "b"], predictors=["p", "q"])
                torch.manual_seed(5)
                m1.train()
                m1.rng_mode = "reference"
                e1, k1 = m1._draw_noise(world * B, True, torch.device("cuda", 0))
                chk("eps = rows of the global draw", np.array_equal(e1.cpu().numpy(), np.concatenate([g["eps"] for g in got])))
                for k in k1:
                    chk(k + " = rows of the global draw",
                        np.array_equal(k1[k].cpu().numpy(), np.concatenate([g["masks"][k] for g in got])))
            if bad:
                ok = False
                print("MISMATCH noise variant, mode", mode, bad)
    _finish(ok, False)


if __name__ == "__main__":
    main()
