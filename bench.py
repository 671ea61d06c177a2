#!/usr/bin/env python3
"""Benchmark of the Celligner2 training step (forward + backward + Adam) on MI355X.

python bench.py --gpus N --steps K --warmup W

N > 1: one rank per GPU over RCCL.  Under torch.distributed.run (WORLD_SIZE set, as the driver launches it) this process
is one of the N ranks; started plainly (`python bench.py --gpus N`) it starts the N ranks itself, as a child
`python -m torch.distributed.run` BEFORE anything touches the GPU, and exits with the child's code.  It never times fewer
GPUs than asked for: a world size that differs from --gpus, or fewer devices than ranks, is an error (exit 2).

Workload (BASELINE.json configs[3], SURVEY.md 8(d) "C4"): synthetic 100k x 19k expression matrix resident in
HBM, hidden [1024,512], latent 64, 3 conditions, NB reconstruction, multi-dataset MMD on z, LayerNorm,
dropout 0.05, Adam(lr 1e-3, eps 0.01, weight_decay 0.04); gene-width GEMMs with bf16 operands / fp32
accumulate; weak scaling, 4096 cells per GPU per step.  A step = exactly Trainer.on_iteration of the reference
(trainer.py:305-323) on a batch gathered from the device-resident dataset.

Prints ONE JSON line (rank 0).  `roofline`: the gene-width MFMA GEMMs (the dominant kernel), timed with events on
their own stream inside the timed region.  `cpu_baseline`: the numpy oracle (oracle/c2_oracle.py, a port) timed
on the host cores on a bounded sample of the same workload.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PMC_JSON = "r03_gemm_pmc.json"  # HBM traffic of the gene-width GEMM launches (tools/pmc_step_json.py)

CONFIGS = {
    # name: G, hidden, latent, C, P, cls_hidden, per-GPU batch, N rows, beta
    "c4": dict(G=19000, hidden=[1024, 512], latent=64, C=3, P=0, cls_hidden=[32], B=4096, N=100000, beta=1.0,
               desc="synthetic 100k x 19k, hidden [1024,512], latent 64, NB + MMD, 3 conditions"),
    "c2": dict(G=19000, hidden=[256, 128], latent=16, C=2, P=30, cls_hidden=[32, 64], B=128, N=11400, beta=500.0,
               desc="CCLE+TCGA shape 11.4k x 19k, hidden [256,128], latent 16, NB + MMD + classifier"),
    "c4z": dict(G=19000, hidden=[1024, 512], latent=64, C=3, P=0, cls_hidden=[32], B=4096, N=100000, beta=1.0, recon="zinb",
                desc="c4 with the ZINB reconstruction (second gene-width head for the dropout logits)"),
    # the other configurations of SURVEY.md 8(d): records kept under profiles/, not the headline
    "c1": dict(G=1000, hidden=[128, 128], latent=10, C=5, P=0, cls_hidden=[32], B=128, N=15681, beta=1.0, recon="mse",
               desc="pancreas shape 15.7k x 1000, hidden [128,128], latent 10, MSE on log1p counts + MMD, 5 conditions"),
    "c3": dict(G=19000, hidden=[820, 512, 256], latent=26, C=3, P=30, cls_hidden=[32, 64], B=4096, N=30000, beta=1.0,
               desc="GTEx+TCGA+CCLE shape 30k x 19k, hidden [820,512,256], latent 26, NB + MMD + classifier, 3 conditions"),
    # gene counts that are not multiples of 8 (rows of the [B, G] operands do not start on 16 bytes unless padded)
    "c4g4": dict(G=19004, hidden=[1024, 512], latent=64, C=3, P=0, cls_hidden=[32], B=4096, N=100000, beta=1.0,
                 desc="c4 with 19 004 genes (a multiple of 4, not of 8)"),
    "c4g1": dict(G=19001, hidden=[1024, 512], latent=64, C=3, P=0, cls_hidden=[32], B=4096, N=100000, beta=1.0,
                 desc="c4 with 19 001 genes (odd)"),
    "c5": dict(G=19000, hidden=[256, 128], latent=16, C=3, P=30, cls_hidden=[32, 64], B=128, N=1000, beta=500.0,
               dr=0.0, surgery=True,
               desc="surgery fine-tune: the c2 model with a third condition, 1000 query rows, freeze + freeze_expression "
                    "(trainable: theta and the condition columns), no dropout"),
}


def synth_dataset(cfg, device, seed=2020):
    """Synthetic NB-like counts generated on the device (there is no network / real matrix)."""
    from celligner2_amd.dataset import DeviceDataset
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    N, G, C, P = cfg["N"], cfg["G"], cfg["C"], cfg["P"]
    gene = torch.exp(torch.randn(1, G, device=device, generator=g) * 1.5 + 1.5)
    X = torch.empty(N, G, device=device)
    good = torch.empty(N, G, device=device, dtype=torch.bool)
    for s in range(0, N, 8192):
        e = min(N, s + 8192)
        depth = torch.empty(e - s, 1, device=device).uniform_(0.5, 2.0, generator=g)
        X[s:e] = torch.poisson(depth * gene, generator=g)
        good[s:e] = torch.rand(e - s, G, device=device, generator=g) > 0.005
    lab = torch.randint(0, C, (N,), device=device, generator=g)
    cond = torch.zeros(N, C, dtype=torch.int64, device=device)
    cond[torch.arange(N, device=device), lab] = 1
    pred = None
    if P > 0:
        pl = torch.randint(0, P, (N,), device=device, generator=g)
        pred = torch.zeros(N, P, dtype=torch.int64, device=device)
        pred[torch.arange(N, device=device), pl] = 1
        pred[torch.rand(N, device=device, generator=g) < 0.2] = -1
    sf = X.sum(1)
    if cfg.get("recon") == "mse":  # the MSE configurations feed log-transformed expression
        X = torch.log1p(X)
    return DeviceDataset.from_tensors(X, sf, good, cond, pred, minweight=0.2, repitch=True)


def _cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(cfg, seconds=25.0, warm=3, timed=20):
    """The numpy oracle's full step (forward + backward + Adam) on a bounded sample of the workload: `warm` warm-up
    steps, then up to `timed` steps (fewer when the time bound is reached), median step time."""
    from oracle import c2_oracle as orc
    Bc = 256
    ocfg = orc.Cfg(input_dim=cfg["G"], hidden_layer_sizes=cfg["hidden"], latent_dim=cfg["latent"],
                   n_conditions=cfg["C"], n_predictors=cfg["P"], classifier_hidden_layer_sizes=cfg["cls_hidden"],
                   recon_loss=cfg.get("recon", "nb"), use_mmd=True, beta=cfg["beta"])
    rng = np.random.default_rng(0)
    params = orc.init_params(ocfg, 0, np.float32)
    x = rng.poisson(rng.uniform(0.5, 2, (Bc, 1)) * np.exp(rng.normal(1.5, 1.5, (1, cfg["G"])))).astype(np.float32)
    cond = np.zeros((Bc, cfg["C"]), np.int64)
    cond[np.arange(Bc), rng.integers(0, cfg["C"], Bc)] = 1
    batch = dict(x=x, batch=cond, sizefactor=x.sum(1), goodloc=np.ones_like(x, dtype=bool))
    if cfg["P"]:
        cls = np.zeros((Bc, cfg["P"]), np.int64)
        cls[np.arange(Bc), rng.integers(0, cfg["P"], Bc)] = 1
        batch["classes"] = cls
    state = {}
    times = []
    t_end = time.time() + seconds
    while len(times) < warm + 2 or (time.time() < t_end and len(times) < warm + timed):
        eps = rng.standard_normal((Bc, cfg["latent"])).astype(np.float32)
        t0 = time.time()
        out = orc.forward_backward(params, ocfg, batch, eps, gscale=(1, 1e-3, 1, 1))
        orc.adam_step(params, out["grads"], state)
        times.append(time.time() - t0)
    kept = times[warm:]
    med = float(np.median(kept))
    cores = int(os.environ.get("OMP_NUM_THREADS", os.cpu_count() or 1))
    return dict(value=Bc / med, unit="cells/s", cores=cores, kind="port", cpu=_cpu_model(),
                sample=f"{warm} warm-up + {len(kept)} timed steps of {Bc} cells (median), same model/genes "
                       f"(numpy fp32 oracle, BLAS threads={cores})")


def _self_launch(n):
    """`python bench.py --gpus N` without a launcher: start the N ranks as a child torch.distributed.run.  Nothing here
    has touched the GPU yet (device_count() does not initialise it), and the child is a fresh process tree."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr",
           "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.call(cmd, env=env)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", default="c4", choices=sorted(CONFIGS))
    ap.add_argument("--precision", default="bf16", choices=["bf16", "fp32"])
    ap.add_argument("--batch", type=int, default=0, help="cells per GPU per step (default: the config's)")
    ap.add_argument("--global-batch", type=int, default=0,
                    help="STRONG scaling: cells per step over all GPUs, each rank takes 1/N (SURVEY.md 8(d): 8192)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--profile-pass", action="store_true",
                    help="warm-up and the K timed steps only (no steady-state loop, no all-terms leg, no CPU baseline): "
                         "what the rocprofv3 passes run, so that dispatch counts divide by W+K")
    ap.add_argument("--graph", type=int, default=-1, help="1/0: replay the step as a HIP graph (default: the config's)")
    ap.add_argument("--steady-seconds", type=float, default=8.0,
                    help="length of the second, longer timing loop (long enough for a 5 s utilisation sampler to see it)")
    args = ap.parse_args()

    backend = os.environ.get("C2_BENCH_BACKEND", "nccl")
    if "WORLD_SIZE" in os.environ or args.gpus == 1:
        # The contract is ONE JSON line on stdout.  Libraries write there too (RCCL prints its version banner to stdout when
        # the first communicator comes up): from here on file descriptor 1 is the error stream, and the result line goes
        # to the saved original.  (The self-launching parent below keeps its stdout: its children inherit it.)
        sys.stdout.flush()
        result_fd = os.dup(1)
        os.dup2(2, 1)
    else:
        result_fd = 1

    def emit(obj):
        os.write(result_fd, (json.dumps(obj) + "\n").encode())

    if "WORLD_SIZE" not in os.environ:
        if args.gpus > 1:
            sys.exit(_self_launch(args.gpus))
        world, rank, local = 1, 0, 0
    else:
        world = int(os.environ["WORLD_SIZE"])
        rank = int(os.environ.get("RANK", "0"))
        local = int(os.environ.get("LOCAL_RANK", "0"))
        if world != args.gpus:
            print(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks", file=sys.stderr)
            sys.exit(2)
    ndev = torch.cuda.device_count()
    if backend == "nccl" and world > ndev:
        print(f"bench.py: {world} ranks over RCCL need {world} GPUs, this node shows {ndev} "
              "(C2_BENCH_BACKEND=gloo rehearses the N-rank code path with ranks sharing a GPU)", file=sys.stderr)
        sys.exit(2)

    import torch.distributed as dist
    if backend != "nccl":
        local %= max(ndev, 1)  # rehearsal: ranks share the GPUs there are
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    comm = None
    # C2_DP_SINGLE_RANK=1 with --gpus 1: the data-parallel pipeline (no graph, bucket exchange on the side stream, sharded
    # Adam, parameter all-gather) with ONE rank over RCCL -- a rehearsal of the N > 1 code path on a one-GPU box; the line
    # it prints says "parallelism": "dp1-rehearsal"
    rehearse = world == 1 and backend == "nccl" and os.environ.get("C2_DP_SINGLE_RANK") == "1"
    if rehearse:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", str(29700 + os.getpid() % 200))
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
    dp = world > 1 or rehearse
    if dp:
        # "nccl" (= RCCL) is the product path; C2_BENCH_BACKEND=gloo lets the N > 1 code path be rehearsed with several
        # ranks sharing one GPU (RCCL needs one device per rank)
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=device)
        else:
            dist.init_process_group(backend)
        if dist.get_world_size() != args.gpus:
            print(f"bench.py: process group has {dist.get_world_size()} ranks, --gpus {args.gpus}", file=sys.stderr)
            sys.exit(2)
        from celligner2_amd.parallel import DataParallelStep
        # reduce-scatter -> Adam on the rank's own chunks -> all-gather (C2_BENCH_SHARD=0: all-reduce, Adam everywhere)
        comm = DataParallelStep(shard_optimizer=os.environ.get("C2_BENCH_SHARD", "1") != "0")

    from celligner2_amd import ops
    from celligner2_amd.model import Celligner2
    from celligner2_amd.trainer import Celligner2Trainer

    cfg = dict(CONFIGS[args.config])
    if args.batch:
        cfg["B"] = args.batch
    if args.global_batch:
        if args.global_batch % world:
            print(f"bench.py: --global-batch {args.global_batch} does not divide over {world} ranks", file=sys.stderr)
            sys.exit(2)
        cfg["B"] = args.global_batch // world
    B = cfg["B"]
    Bglob = B * world
    # every batch of an epoch is a full global batch: the loader keeps a short last batch (as the reference's does), which
    # would put a first-seen shape -- and its buffer allocation -- inside the timed region
    n_samples = (cfg["N"] // Bglob) * Bglob
    if n_samples == 0:
        print(f"bench.py: global batch {Bglob} exceeds the dataset ({cfg['N']} rows)", file=sys.stderr)
        sys.exit(2)
    torch.manual_seed(2020)
    recon = cfg.get("recon", "nb")
    model = Celligner2(cfg["G"], cfg["hidden"], cfg["latent"], cfg.get("dr", 0.05), False, recon,
                       conditions=[f"c{i}" for i in range(cfg["C"])],
                       use_mmd=True, mmd_on="z", beta=cfg["beta"], predictors=[f"p{i}" for i in range(cfg["P"])],
                       classifier_hidden_layer_sizes=cfg["cls_hidden"], precision=args.precision)
    if cfg.get("surgery"):
        from celligner2_amd.surgery import apply_freeze
        apply_freeze(model.to(device), freeze_expression=True)
    data = synth_dataset(cfg, device)
    # the step is replayed as a HIP graph unless --graph 0 -- in a data-parallel run over RCCL too (its collectives are
    # captured with the kernels; a host-side gloo rehearsal cannot be).  If the collective library refuses the capture the
    # trainer goes on launching the step kernel by kernel and the line says so ("graph_refused").
    use_graph = (bool(cfg.get("graph", True)) if args.graph < 0 else bool(args.graph)) and (not dp or backend == "nccl")
    trainer = Celligner2Trainer(model, None, train_data=data, valid_data=None, batch_size=Bglob, use_leiden=False,
                                alpha_kl=1e-3, monitor=False, use_early_stopping=False, rng="device", world=comm,
                                seed=2020, n_samples=n_samples, **({"graph": True} if use_graph else {}))
    trainer.initialize_loaders()
    model.train()
    trainer.optimizer = trainer.make_optimizer(lr=1e-3, eps=0.01)
    trainer.epoch, trainer.iter = 0, 0
    from collections import defaultdict
    trainer.iter_logs = defaultdict(list)

    def batches():
        while True:
            for b in trainer.dataloader_train:
                yield b

    stream = batches()
    rows_done = [0]

    def step(src):
        b = next(src)
        # rows this rank really processed (host-side shape, no sync); a captured step's batch carries its row count
        rows_done[0] += int(b.n) if hasattr(b, "n") else int(b["sizefactor"].numel())
        trainer.on_iteration(b)

    def sync():
        if dp:
            dist.barrier()
        torch.cuda.synchronize()

    t_start = time.perf_counter()

    def trace(msg):  # C2_BENCH_TRACE=1: stage markers on stderr (rehearsals of the multi-rank path)
        if os.environ.get("C2_BENCH_TRACE"):
            import faulthandler
            faulthandler.dump_traceback_later(60, exit=True)  # re-armed at every marker: fires only on a stall
            print(f"[bench rank {rank} t={time.perf_counter() - t_start:9.4f}] {msg}", file=sys.stderr, flush=True)

    def timed(n, src, host=False):
        """n steps between two (barrier + device sync) points: (seconds, rows of this rank, host launch seconds)."""
        sync()
        rows_done[0] = 0
        t0 = time.perf_counter()
        for i in range(n):
            step(src)
            trace(f"timed step {i} launched")
        t_host = time.perf_counter() - t0  # the launch path alone (the GPU queue drains in sync())
        sync()
        return time.perf_counter() - t0, rows_done[0], t_host

    def over_ranks(dt, rows):
        """MAX of the time, SUM of the rows over ranks."""
        if not dp:
            return dt, rows
        t = torch.tensor([dt], device=device, dtype=torch.float64)
        r = torch.tensor([rows], device=device, dtype=torch.int64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dist.all_reduce(r, op=dist.ReduceOp.SUM)
        return float(t.item()), int(r.item())

    trace("model, dataset and trainer built")
    # The objects built so far (modules, the dataset's tensors, everything imported) are permanent: keep the cyclic
    # collector from walking them in the loop.  A step launched kernel by kernel creates a few thousand short-lived Python
    # objects, and a full collection over the whole heap -- measured as a 70-100 ms stall of the launch thread once in
    # ~25 steps of the data-parallel pipeline -- costs more than 30 steps.
    import gc
    gc.collect()
    gc.freeze()
    if use_graph:  # set-up, like building the model: two ordinary steps, then the step that captures the graph
        for _ in range(3):
            step(stream)
    for i in range(args.warmup):
        step(stream)
        trace(f"warmup step {i} launched")
    sync()
    trace("warmup done")
    # (flops, start_event, end_event) of every gene-width GEMM launch in the timed region.  A replayed graph cannot carry
    # the events: with --graph the same K steps are timed once more further down, launched the ordinary way, for them.
    ops.PROFILE_GEMM = None if use_graph else []
    prof_host = None
    if os.environ.get("C2_BENCH_CPROFILE"):  # developer aid: cProfile of the host launch path of the timed steps
        import cProfile
        prof_host = cProfile.Profile()
        prof_host.enable()
    dt, rows, t_host = timed(args.steps, stream)
    if prof_host is not None:
        import pstats
        prof_host.disable()
        pstats.Stats(prof_host, stream=sys.stderr).sort_stats("tottime").print_stats(35)
    prof, ops.PROFILE_GEMM = ops.PROFILE_GEMM, None
    losses_seen = [float(trainer.current_loss.detach())]
    trace("timed steps done")
    model._engine().check_status()
    dt, cells = over_ranks(dt, rows)
    # The same pipeline over a longer loop (>= --steady-seconds): the K-step region above lasts tens of milliseconds,
    # too short for a power-managed chip to settle; `value` stays the K-step figure the contract asks for.
    if args.profile_pass:
        if rank == 0:
            emit({"profile_pass": True, "steps": args.steps, "warmup": args.warmup,
                  "hip_graph": use_graph and trainer.graph_refused is None,
                              "ms_per_step": 1e3 * dt / args.steps, "value": cells / dt})
        if dp:
            dist.destroy_process_group()
        return
    n_steady = max(args.steps, int(np.ceil(args.steady_seconds / max(dt / args.steps, 1e-6))))
    if dp:  # every rank must run the same number of steps
        ns = torch.tensor([n_steady], device=device, dtype=torch.int64)
        dist.all_reduce(ns, op=dist.ReduceOp.MAX)
        n_steady = int(ns.item())
    dt_ss, rows_ss, _ = timed(n_steady, stream)
    dt_ss, cells_ss = over_ranks(dt_ss, rows_ss)
    trace("steady-state loop done")
    # Third, untuned-pipeline timing of the same K steps: every data-only term is evaluated INSIDE the step, as the
    # reference's forward does (log(1+x) of the batch, lgamma(x+1) of the NB likelihood; the batch's counts and mask
    # copied by the gather) instead of once per resident dataset.  Reported beside the headline value.
    graph_refused = trainer.graph_refused
    if use_graph:
        trainer.drop_graph()
        trainer.graph = trainer.dataloader_train.static = False
        eager = batches()
        for _ in range(2):
            step(eager)
        ops.PROFILE_GEMM = []
        timed(args.steps, eager)
        prof, ops.PROFILE_GEMM = ops.PROFILE_GEMM, None
    all_in_step = None
    if not dp and recon == "nb":  # (single GPU only: the scaling runs time the headline pipeline and nothing else)
        ld = trainer.dataloader_train
        ld.nb_const, ld.resident_xin = False, (lambda n: False)
        stream2 = batches()
        for _ in range(2):
            step(stream2)
        dt2, rows2, _ = timed(args.steps, stream2)
        all_in_step = {"value": rows2 / dt2, "unit": "cells/s", "ms_per_step": 1e3 * dt2 / args.steps}
    trace("second timing done")
    if os.environ.get("C2_BENCH_TRACE"):
        import faulthandler
        faulthandler.cancel_dump_traceback_later()
    last = losses_seen + [float(v[-1]) for k, v in sorted(trainer.iter_logs.items())]
    assert all(np.isfinite(last)), f"non-finite losses {last}"
    n_gpus = dist.get_world_size() if dp else 1

    if rank == 0:
        gemm_ms = sum(s.elapsed_time(e) for _, s, e in prof)
        gemm_flops = sum(f for f, _, _ in prof)
        n_launch = max(len(prof), 1)
        peak = 2500.0 if args.precision == "bf16" else 157.3
        achieved = gemm_flops / (gemm_ms * 1e-3) / 1e12 if gemm_ms > 0 else 0.0
        # HBM bytes per GEMM launch: measured in separate rocprofv3 --pmc passes (FETCH_SIZE x2 + WRITE_SIZE, see
        # profiles/*_gemm_pmc.json); only valid for the configuration it was measured on
        traffic, traffic_note = None, None
        pmc = os.path.join(ROOT, "profiles", PMC_JSON)
        if args.config == "c4" and args.precision == "bf16" and B == CONFIGS["c4"]["B"] and os.path.exists(pmc):
            rec = json.load(open(pmc))
            # the counter passes must have seen the launches this run timed: same number of gene-width launches per step
            if abs(rec["launches_per_step"] - n_launch / args.steps) < 1e-6:
                traffic = rec["traffic_bytes_per_launch"]
            else:
                traffic_note = (f"profiles/{PMC_JSON} holds {rec['launches_per_step']} launches per step, this run timed "
                                f"{n_launch / args.steps}: not the same launch set, traffic withheld")
        H1 = cfg["hidden"][0]
        out = {
            "metric": "training cells/sec (fwd+bwd+Adam) at 19k genes" if cfg["G"] == 19000 else
                      f"training cells/sec (fwd+bwd+Adam) at {cfg['G']} genes",
            "value": cells / dt, "unit": "cells/s", "n_gpus": n_gpus, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * dt / args.steps, "host_ms_per_step": 1e3 * t_host / args.steps,
            "higher_is_better": True, "scaling": "strong" if args.global_batch else "weak", "vs_baseline": None,
            "dtype": "bf16" if args.precision == "bf16" else "f32", "data": "synthetic",
            "cells_timed": cells,
            "config": {"workload": f"{args.config}: {cfg['desc']}; {B} cells/GPU/step, global batch {Bglob}, "
                                   f"dataset resident in HBM, {recon.upper()} reconstruction, gene-width GEMMs {args.precision} operands "
                                   f"fp32 accumulate",
                       "global_batch": Bglob, "parallelism": f"dp{n_gpus}" + ("-rehearsal" if rehearse else ""), "backend": backend if dp else None,
                       "epoch_rows": n_samples, "hip_graph": use_graph and graph_refused is None,
                       "graph_refused": graph_refused,
                       "optimizer": ("sharded over ranks (reduce-scatter, Adam on own chunks, all-gather)"
                                     if (comm is not None and comm.sharded) else "every rank updates every parameter"),
                       "dataset_resident_terms": "log(1+x) as bf16 and the per-row sum of lgamma(x+1) are evaluated once "
                                                 "per dataset (identical values); see all_terms_in_step for the step "
                                                 "that evaluates them per batch like the reference's forward"},
            "steady_state": {"value": cells_ss / dt_ss, "unit": "cells/s", "steps": n_steady,
                             "ms_per_step": 1e3 * dt_ss / n_steady, "seconds": dt_ss},
            "all_terms_in_step": all_in_step,
            "roofline": {"bound": "mfma", "achieved": achieved, "peak": peak, "unit": "TFLOP/s",
                         "frac": achieved / peak, "traffic": traffic, "traffic_note": traffic_note,
                         "kernel": "c2::wide::gemm_wide_kernel<*,*,8,*> + c2::wide::gemm_wide16_kernel<*,*,*> "
                                   "(gene-width Linear fwd / dgrad / wgrad)",
                         "timed_in": "the K timed steps" if not use_graph else "the same K steps launched without the graph",
                         "launches_per_step": n_launch / args.steps, "avg_launch_ms": gemm_ms / n_launch,
                         "gemm_ms_per_step": gemm_ms / args.steps, "gemm_flops_per_step": gemm_flops / args.steps,
                         # (gene-width = a weight dimension >= 2048; the frozen surgery step has no gene-width weight gradient)
                         "gemm_flops_per_step_expected": ((3 if cfg.get("surgery") else 5) * 2.0 * B * cfg["G"] * H1
                                                          if cfg["G"] >= 2048 else 0.0)},
        }
        if not args.no_cpu_baseline and not dp:
            out["cpu_baseline"] = cpu_baseline(cfg)
        emit(out)
    if dp:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
