"""Drop-in `Trainer` / `Celligner2Trainer` / `EarlyStopping` for the MI355X step.

Public surface = the reference's (celligner2/trainers/celligner2/trainer.py:66-370, semisupervised.py:55-74,
utils/monitor.py:32-115): constructor keywords, `train(n_epochs, lr, eps)`, the overridable hooks
(`before_loop`, `on_epoch_begin`, `on_iteration`, `loss`, `on_epoch_end`, `after_loop`), the public state
(`logs`, `iter_logs`, `epoch`, `iter`, `optimizer`, `best_epoch`, `training_time`, `dataloader_train/valid`,
`train_data/valid_data`) and the `logs` keys.  Behaviour is pinned to the reference by the `trainer_*` fixtures
(tools/gen_golden_trainer.py -> tests/test_trainer_gpu.py): same rows visited, same draws, same logs and weights.

What is different underneath: the dataset is resident in HBM and a batch is gathered by row index on the device
(`_Loader`); loss -> zero_grad -> backward -> clip -> Adam runs in the HIP library (clipping inside the fused Adam
kernel); per-iteration loss scalars stay on the device and are averaged once per epoch (`log_sync="epoch"`) instead of
six `.item()` syncs per step (`log_sync="step"` reproduces the reference's per-step floats and error timing).

`graph=True` (rng="device"): after two ordinary steps the whole iteration -- gather of the batch's rows,
forward, backward, Adam -- is captured once as a HIP graph and every further full-size batch is ONE graph launch
(`_StepGraph`); what changes between steps (row indices, KL weight, learning rate, step count, dropout counter, loss
sums) lives in device memory.  Short last batches and validation run the ordinary way.  With `world` (one process per
GPU over RCCL) the captured iteration includes the collectives -- latent all-gather, bucket reduce-scatter / all-reduce
on the side stream, sharded Adam, parameter all-gather: RCCL's calls are stream-ordered and capturable, so the
data-parallel step costs the host what the single-GPU step costs (one graph launch).
"""
from __future__ import annotations

import os
import sys
import time
from collections import defaultdict

import numpy as np
import torch

from . import ops
from .dataset import make_dataset
from .optim import FusedAdam

_INF = float("inf")


class EarlyStopping(object):
    """Plateau monitor (utils/monitor.py:32-115): after every epoch `step(value)` says whether to go on and whether to
    decay the learning rate; `update_state(value)` says whether `value` is the best seen (for the best-state snapshot).

    State machine of `step`, epochs counted from 1:
      grace     epoch < patience            -> (continue, no decay); NOTHING is tracked, not even the best value
      spent     wait >= patience            -> (stop, no decay)
      tracking  otherwise                   -> decay iff reduce_lr and wait_lr >= lr_patience (wait_lr restarts),
                                               then both counters grow by one when the gain over the best value is
                                               below `threshold`, else restart; the best value moves on any gain > 0
    `patience == 0` disables the monitor the way the reference does: `step` then returns a bare False."""

    def __init__(self, early_stopping_metric: str = "val_unweighted_loss", mode: str = "min", threshold: float = 0,
                 patience: int = 20, reduce_lr: bool = True, lr_patience: int = 13, lr_factor: float = 0.1):
        self.early_stopping_metric, self.mode, self.threshold = early_stopping_metric, mode, threshold
        self.patience, self.reduce_lr, self.lr_patience, self.lr_factor = patience, reduce_lr, lr_patience, lr_factor
        self._toward = 1.0 if mode == "min" else -1.0  # gain(best, value) = toward * (best - value)
        self.epoch = self.wait = self.wait_lr = 0
        self.current_performance = _INF
        self.best_performance = self.best_performance_state = self._toward * _INF

    def _gain(self, best, value):
        return self._toward * (best - value)

    def step(self, scalar):
        if self.patience == 0:
            return False
        self.epoch += 1
        if self.epoch < self.patience:
            return True, False
        if self.wait >= self.patience:
            print(f"\nEarly stop: {self.patience} epochs without a gain of {self.threshold} or more in "
                  f"{self.early_stopping_metric}; loosen early_stopping_kwargs if that is too strict.")
            return False, False
        decay = bool(self.reduce_lr) and self.wait_lr >= self.lr_patience
        if decay:
            self.wait_lr = 0
        self.current_performance = scalar
        gain = self._gain(self.best_performance, scalar)
        if gain > 0:
            self.best_performance = scalar
        stalled = gain < self.threshold
        self.wait = self.wait + 1 if stalled else 0
        self.wait_lr = self.wait_lr + 1 if stalled else 0
        return True, decay

    def update_state(self, scalar):
        better = self._gain(self.best_performance_state, scalar) > 0
        if better:
            self.best_performance_state = scalar
        return better


def print_progress(epoch, logs, n_epochs=10000, only_val_losses=True):
    """One-line progress report (the reference prints a bar plus the last value of every log key that is not an
    `unweighted` one, validation keys only by default: trainers/celligner2/_utils.py:13-76)."""
    shown = [k for k in logs if "unweighted" not in k and (not only_val_losses or "val_" in k)]
    done = epoch + 1
    width = 20
    fill = int(width * done // n_epochs)
    line = "\r |{}{}| {:.1f}% ".format("█" * fill, "-" * (width - fill), 100.0 * done / float(n_epochs))
    line += "".join(f" - {k:s}: {logs[k][-1]:7.10f}" for k in shown)
    sys.stdout.write(line + ("\n" if done == n_epochs else ""))
    sys.stdout.flush()


class _StaticBatch:
    """What a loader in static mode yields for a full-size batch: the rows are already in `loader.static_rows` (a
    persistent device buffer a captured step reads); `loader.gather_static()` materialises the batch dict."""

    def __init__(self, loader, n):
        self.loader, self.n = loader, n


class _Loader:
    """Index stream of the reference's DataLoader(+WeightedRandomSampler / shuffle) served from a DeviceDataset.

    rng="reference" consumes torch's CPU generator exactly as torch.utils.data does (one int64 `random_()` per
    iterator, then `multinomial` / a re-seeded `randperm`), so a seeded run visits the same rows as the reference;
    rng="device" draws on the GPU generator.  Like the reference's loader the last batch of an epoch may be short.

    Data-parallel runs (`world`): `batch_size` is the GLOBAL batch; every rank draws the same index stream and keeps
    its own contiguous share of each global batch.  Shares are equal (the MMD all-gather needs that), so a short last
    batch loses its last `len % world_size` rows and a last batch with fewer rows than ranks is skipped."""

    def __init__(self, dataset, batch_size, weights=None, num_samples=None, rng="reference", xin_dtype=None,
                 apply_log=False, world=None, nb_const=False, resident_xin=None, centered=False, xin2=False):
        self.centered = centered  # bf16 encoder input minus the dataset's per-gene mean (DeviceDataset.xin_center)
        self.xin2 = xin2          # an IEEE-half encoder input travels with its bf16 twin (DeviceDataset.gather)
        self.nb_const = nb_const  # deliver the data-only NB term (sum goodloc*lgamma(x+1)) with every batch
        # callable(batch rows) -> bool: leave the encoder input in the dataset's resident bf16 matrix (read in place by the
        # row-indexed first-layer GEMMs) instead of writing a per-batch copy
        self.resident_xin = resident_xin or (lambda n: False)
        self.ds, self.bs, self.weights, self.num_samples, self.rng = dataset, batch_size, weights, num_samples, rng
        self.xin_dtype, self.apply_log = xin_dtype, apply_log
        self.world = world
        # rows of a full batch that this process handles: the batch, or this rank's share of the global batch
        self.share = batch_size // world.world_size if (world is not None and world.world_size > 1) else batch_size
        self._bufs = {}
        self.static = False      # full-size batches are handed over as _StaticBatch (Trainer(graph=True))
        self.static_rows = None  # int64 [batch_size] on the device

    def gather_static(self):
        n = self.static_rows.numel()
        return self.ds.gather(self.static_rows, self.xin_dtype, self.apply_log, self._bufs.setdefault(("static", n), {}),
                              self.nb_const, resident_rows=self.nb_const, resident_xin=self.resident_xin(n),
                              centered=self.centered, xin2=self.xin2)

    def __len__(self):
        n = self.num_samples if self.weights is not None else len(self.ds)
        return int(np.ceil(n / self.bs))

    def _indices(self):
        dev = self.ds.device
        if self.rng == "reference":
            torch.empty((), dtype=torch.int64).random_()  # _BaseDataLoaderIter base seed
            if self.weights is not None:
                idx = torch.multinomial(self.weights, self.num_samples, True)
            else:
                seed = int(torch.empty((), dtype=torch.int64).random_().item())
                g = torch.Generator()
                g.manual_seed(seed)
                idx = torch.randperm(len(self.ds), generator=g)
            return idx.to(dev)
        if self.weights is not None:
            # (the weights go to the device once: a copy from pageable host memory per epoch would hold the launch thread
            # until the GPU has drained everything queued before it)
            # Sampling with replacement = inverting the cumulative distribution at uniform draws.  The weights are fixed,
            # so their normalised cumulative sum (float64, as torch.multinomial normalises) is formed once and an epoch's
            # index stream costs two launches instead of torch.multinomial's dozen (0.3 ms of the GPU's time per epoch
            # boundary at 100 000 rows; same distribution, its own stream of draws).
            cdf = self.__dict__.get("_weights_cdf")
            if cdf is None or cdf.device != dev:
                w = self.weights.to(dev).double()
                if not bool((w >= 0).all()) or not bool(torch.isfinite(w).all()) or float(w.sum()) <= 0:
                    raise RuntimeError("invalid sampling weights (negative, non-finite or all zero)")
                cdf = self.__dict__["_weights_cdf"] = torch.cumsum(w / w.sum(), 0)
                cdf[-1] = 1.0
            u = torch.rand(self.num_samples, device=dev, dtype=torch.float64)
            return torch.searchsorted(cdf, u, right=True).clamp_(max=cdf.numel() - 1)
        return torch.randperm(len(self.ds), device=dev)

    def __iter__(self):
        idx = self._indices()
        for s in range(0, idx.numel(), self.bs):
            # one buffer set per batch shape: the short last batch must not evict the full-size buffers
            sub = idx[s:s + self.bs]
            if self.world is not None and self.world.world_size > 1:
                per = sub.numel() // self.world.world_size
                if per == 0:
                    continue
                sub = sub[self.world.rank * per:(self.world.rank + 1) * per]
            if self.static and sub.numel() == self.share:
                if self.static_rows is None:
                    self.static_rows = torch.empty(self.share, dtype=torch.int64, device=idx.device)
                self.static_rows.copy_(sub)
                yield _StaticBatch(self, self.share)
                continue
            yield self.ds.gather(sub, self.xin_dtype, self.apply_log, self._bufs.setdefault(sub.numel(), {}),
                                 self.nb_const, resident_rows=self.nb_const, resident_xin=self.resident_xin(sub.numel()),
                                 centered=self.centered, xin2=self.xin2)


# keyword -> default of the options the reference pops from **kwargs (trainer.py:117-137), then the build-only ones
_POPPED = (("alpha_iter_anneal", None), ("n_samples", None), ("train_frac", 0.7), ("use_stratified_sampling", True),
           ("weight_decay", 0.04), ("clip_value", 0.0), ("n_workers", 0), ("seed", 2020), ("monitor", True),
           ("monitor_only_val", True),
           ("log_sync", "epoch"),   # "step": python floats every iteration (and device errors raised at once)
           ("rng", "reference"),    # index stream / noise on torch's CPU generator, in the reference's order
           ("world", None),         # parallel.DataParallelStep of a one-process-per-GPU run
           ("graph", False))        # replay the step as a captured HIP graph (fixed-shape batches only)


class _StepGraph:
    """One training iteration as a HIP graph.

    The first WARMUP calls run the iteration eagerly (every workspace gets allocated, the optimizer state and the
    operand copies reach their steady state), the next call captures it, and from then on `step()` is: write what
    changed on the host into device memory (KL weight, learning rate -- only when they changed), launch the graph,
    do the optimizer's host-side bookkeeping.  The six loss scalars of every step are summed on the device (float64)
    and read once per epoch by `drain()`.

    Valid while the model keeps its engine (no parameter replaced, frozen, moved), stays in training mode and the
    optimizer is the one captured: `alive()` checks that, and the trainer drops the graph otherwise."""

    WARMUP = 2

    def __init__(self, trainer):
        self.trainer, self.graph, self.calls, self.count = trainer, None, 0, 0
        self.refused = None  # why the capture was given up (data-parallel only), else None
        self.raw = None      # hipGraphExec_t of the captured step
        dev = trainer.device
        self.alpha = trainer.calc_alpha_coeff()
        self.weights = torch.tensor([1.0, self.alpha, 1.0, 1.0], dtype=torch.float32, device=dev)
        self.sums = torch.zeros(6, dtype=torch.float64, device=dev)
        self.unit = torch.ones((), device=dev)
        self.optimizer = trainer.optimizer
        self.engine = None
        self.total = None
        self.scratch = None
        trainer.model.loss_sums(self.sums)
        trainer.model.device_noise_counter(True)

    def _iteration(self):
        t = self.trainer
        batch = t.dataloader_train.gather_static()
        total, _ = t.model.forward_total(self.weights, **batch)
        t.optimizer.zero_grad()
        total.backward(gradient=self.unit)
        t.optimizer.step()
        return total

    def alive(self):
        t = self.trainer
        return (t.optimizer is self.optimizer and t.model.training
                and (self.engine is None or t.model.__dict__.get("_eng") is self.engine))

    def step(self):
        t = self.trainer
        alpha = t.calc_alpha_coeff()
        if alpha != self.alpha:
            self.alpha = alpha
            self.weights[1:2].fill_(alpha)
        if self.graph is None and self.calls >= self.WARMUP and not self.refused:
            torch.cuda.synchronize()
            graph = torch.cuda.CUDAGraph()
            engine = t.model._engine()
            engine.recorded = self.scratch = []  # the graph owns every scratch tensor its kernels address
            # (data-parallel: the process group's watchdog thread makes HIP calls of its own while this thread captures --
            # only this thread's calls belong to the capture)
            mode = {"capture_error_mode": "thread_local"} if t.world is not None else {}
            if t.world is not None:
                # ... and it must find nothing to do meanwhile: the collectives of the warm-up steps are complete (the
                # synchronize above) but stay on the watchdog's list until its next pass (every 100 ms), and a pass that
                # retires them while this thread captures queries and destroys their events inside the capture window --
                # seen once in ~25 captures as an exception in the watchdog thread, which ends the process (SIGABRT).
                # Two and a half of its periods, once per training run.
                time.sleep(0.25)
            steps = t.optimizer.step_counts()
            try:
                with torch.cuda.graph(graph, **mode):
                    self.total = self._iteration()
            except Exception as exc:  # noqa: BLE001
                if t.world is None:
                    raise
                # A collective library that refuses stream capture: nothing of the attempt has run.  Take the host-side
                # bookkeeping of the attempt back and go on launching the step the ordinary way (recorded in
                # `self.refused`, which bench.py reports).
                self.refused = t.graph_refused = f"{type(exc).__name__}: {exc}"
                t.optimizer.restore_step_counts(steps)
                t.world.forget_pending()
                torch.cuda.synchronize()
                graph = None
            finally:
                engine.recorded = None
            if graph is not None:
                t.optimizer.captured()  # recorded, not run: the replay below performs this step
                self.graph, self.engine = graph, engine
                # The step consumes no torch generator (rng="device": its noise is c2_noise_dev's, counter in device memory),
                # so a replay is hipGraphLaunch and nothing else; CUDAGraph.replay() first refreshes the seed / offset
                # tensors of every generator registered with the capture -- two fill launches per step that nothing reads.
                # (C2_GRAPH_RAW=0: replay through torch.)
                self.raw = graph.raw_cuda_graph_exec() if os.environ.get("C2_GRAPH_RAW") != "0" else None
        if self.graph is not None:
            t.optimizer.sync_hyper()
            if self.raw:
                ops.graph_launch(self.raw)
            else:
                self.graph.replay()
            t.optimizer.replayed()
            t.graph_replays += 1
        else:
            self.total = self._iteration()
        self.calls += 1
        self.count += 1
        return self.total

    def drain(self):
        """(sums [6] as python floats, number of steps) since the last call; one host read."""
        if self.count == 0:
            return None, 0
        sums, n = self.sums.tolist(), self.count
        self.sums.zero_()
        self.count = 0
        return sums, n

    def close(self):
        model = self.trainer.model
        model.loss_sums(None)
        model.device_noise_counter(False)
        self.graph = self.raw = None


class Trainer:
    def __init__(self, model, adata, condition_keys: list = None, predictor_keys: list = None,
                 cell_type_key: str = None, batch_size: int = 128, alpha_epoch_anneal: int = None,
                 alpha_kl: float = 0.001, use_early_stopping: bool = True, reload_best: bool = True,
                 early_stopping_kwargs: dict = None, min_weight: float = 0.2, use_leiden: bool = True,
                 leiden_neighbors: int = 8, leiden_resolution: float = 1, dataset_key="dataset", **kwargs):
        if cell_type_key is None and use_leiden:
            raise NotImplementedError(
                "use_leiden=True needs scanpy's Leiden clustering (trainer.py:91-113), which is data preparation "
                "outside the accelerated path: pass cell_type_key=<obs column> or use_leiden=False")
        if not torch.cuda.is_available():
            raise RuntimeError("celligner2_amd needs an MI355X GPU: there is no CPU training path")
        self.model, self.adata = model, adata
        self.condition_keys, self.predictor_keys, self.cell_type_key = condition_keys, predictor_keys, cell_type_key
        self.batch_size, self.min_weight = batch_size, min_weight
        self.alpha_kl, self.alpha_epoch_anneal = alpha_kl, alpha_epoch_anneal
        self.use_early_stopping, self.reload_best = use_early_stopping, reload_best
        for key, default in _POPPED:
            setattr(self, key, kwargs.pop(key, default))
        staged = (kwargs.pop("train_data"), kwargs.pop("valid_data", None)) if "train_data" in kwargs else None
        self.early_stopping = EarlyStopping(**(early_stopping_kwargs or {}))
        # run state
        self.epoch, self.iter, self.n_epochs = -1, 0, None
        self.best_epoch = self.best_state_dict = self.current_loss = self.optimizer = self.sampler = None
        self.previous_loss_was_nan, self.nan_counter, self.training_time = False, 0, 0
        self.dataloader_train = self.dataloader_valid = self.iters_per_epoch = self.val_iters_per_epoch = None
        self.logs = defaultdict(list)
        self.iter_logs = defaultdict(list)
        # device, seeds (trainer.py:141-145), data-parallel wiring
        torch.manual_seed(self.seed)
        torch.cuda.manual_seed(self.seed)
        self.device = torch.device("cuda", torch.cuda.current_device())
        self.model.to(self.device)
        self.model.rng_mode = self.rng
        self._graph = None
        self.graph_replays, self.graph_refused = 0, None  # steps replayed from a captured graph; why a capture was given up
        if self.graph:
            if self.rng != "device" or self.log_sync != "epoch":
                raise ValueError("graph=True replays the step with device-side noise and per-epoch logs: "
                                 "it needs rng='device' and log_sync='epoch'")
            if self.world is not None and getattr(self.world, "_host_backend", False):
                raise ValueError("graph=True with a data-parallel `world` needs the RCCL backend (\"nccl\"): a host-side "
                                 "exchange (gloo) cannot be captured")
        if self.world is not None:
            self.world.attach_model(self.model)
        if staged is not None:  # DeviceDataset objects staged by the caller (adata may be None)
            self.train_data, self.valid_data = staged
        else:
            self.train_data, self.valid_data = make_dataset(
                adata, train_frac=self.train_frac, condition_keys=condition_keys, cell_type_key=cell_type_key,
                condition_encoder=model.condition_encoder, cell_type_encoder=model.cell_type_encoder,
                predictor_keys=predictor_keys, predictor_encoder=model.predictor_encoder, min_weight=min_weight,
                device=self.device)

    # ------------------------------------------------------------------ loaders (trainer.py:184-229)
    def _loader(self, data, batch_size, weights, n_samples):
        model = self.model
        eng = model._engine()
        xin_dtype = eng.fwd16 or eng.enc[0].gt
        log_in = model.applylog and model.recon_loss != "mse"
        if not log_in and xin_dtype == torch.float32:
            xin_dtype = None  # the encoder reads x itself
        # NB and ZINB read the batch's counts and mask where the dataset lives and take the data-only term
        # sum goodloc * lgamma(x + 1) per row from the dataset (lgamma(1) = 0: the same constant serves both)
        nb = model.recon_loss in ("nb", "zinb")
        G, H0 = model.input_dim, model.hidden_layer_sizes[0]

        def in_place(n):  # both first-layer GEMMs (forward [n,H0] over G, weight gradient [H0,G] over n) on the 256x256 kernel
            from . import ops
            return (nb and xin_dtype in (torch.bfloat16, torch.float16) and n % 8 == 0
                    and ops.gemm_tile(torch.bfloat16, n, H0, ops.pitch2(G)) == 256  # (reduced over the rows' padded length)
                    and ops.gemm_tile(torch.bfloat16, H0, G, n) == 256)

        return _Loader(data, batch_size, weights, n_samples, self.rng, xin_dtype, log_in, self.world, nb, in_place,
                       centered=eng.center_input, xin2=eng.fwd16 is not None)

    def initialize_loaders(self):
        n_train = len(self.train_data)
        if self.n_samples is None or self.n_samples > n_train:
            self.n_samples = n_train
        self.iters_per_epoch = int(np.ceil(self.n_samples / self.batch_size))
        if self.use_stratified_sampling:  # WeightedRandomSampler(weights, n_samples, replacement=True)
            self.dataloader_train = self._loader(self.train_data, self.batch_size,
                                                 torch.tensor(self.train_data.stratifier_weights), self.n_samples)
        else:                             # shuffle=True
            self.dataloader_train = self._loader(self.train_data, self.batch_size, None, None)
        self.dataloader_train.static = bool(self.graph)
        if self.valid_data is not None:
            self.val_iters_per_epoch = int(np.ceil(len(self.valid_data) / self.batch_size))
            self.dataloader_valid = self._loader(self.valid_data, min(self.batch_size, len(self.valid_data)), None, None)

    def make_optimizer(self, lr=1e-3, eps=0.01):
        """trainer.py:262-266: Adam over the parameters that require a gradient, L2 weight decay; here the fused
        multi-tensor kernel, wired to the engine's operand copies and (data-parallel runs) to the gradient exchange."""
        lookup, updated = self.model._adam_hooks()
        return FusedAdam([p for p in self.model.parameters() if p.requires_grad], lr=lr, eps=eps,
                         weight_decay=self.weight_decay, clip_value=self.clip_value, shadow_lookup=lookup,
                         on_updated=updated, grad_sync=self.world, capturable=bool(self.graph))

    def drop_graph(self):
        """Forget the captured step (the next full-size batch starts a new warm-up and capture); the loss sums it
        holds are folded into the current epoch's logs first."""
        if self._graph is not None:
            self._bank_graph_sums()
            self._graph.close()
            self._graph = None

    def calc_alpha_coeff(self):
        """KL weight in force (trainer.py:231-253): linear ramp over epochs or over iterations up to alpha_kl."""
        if self.alpha_epoch_anneal is not None:
            progress = self.epoch / self.alpha_epoch_anneal
        elif self.alpha_iter_anneal is not None:
            progress = (self.epoch * self.iters_per_epoch + self.iter) / self.alpha_iter_anneal
        else:
            return self.alpha_kl
        return min(self.alpha_kl * progress, self.alpha_kl)

    # ------------------------------------------------------------------ the loop (trainer.py:255-294)
    def train(self, n_epochs: int = 400, lr: float = 1e-3, eps: float = 0.01):
        self.initialize_loaders()
        started = time.time()
        self.model.train()
        self.n_epochs = n_epochs
        self.drop_graph()
        self.optimizer = self.make_optimizer(lr, eps)
        self.before_loop()
        for self.epoch in range(n_epochs):
            self.on_epoch_begin(lr, eps)
            self.iter_logs = defaultdict(list)
            for self.iter, batch in enumerate(self.dataloader_train):
                self.on_iteration(batch)
            self.on_epoch_end()
            if self.use_early_stopping and not self.check_early_stop():
                break
        if self.reload_best and self.best_state_dict is not None:
            # (the snapshot is the live state_dict, as in the reference: trainer.py:359 keeps no copy)
            print(f"Reloading the best state (epoch {self.best_epoch}) into the network")
            self.model.load_state_dict(self.best_state_dict)
        self.drop_graph()
        self.model.eval()
        self.after_loop()
        self.model._engine().check_status()
        self.training_time += time.time() - started

    def before_loop(self):
        pass

    def on_epoch_begin(self, lr, eps):
        pass

    def after_loop(self):
        pass

    def _unhook_frozen_batchnorm(self):
        """trainer.py:307-312: BatchNorm layers whose affine parameters surgery froze stop tracking running statistics."""
        for module in self.model.modules():
            if isinstance(module, torch.nn.BatchNorm1d) and not module.weight.requires_grad:
                module.affine = module.track_running_stats = False

    def on_iteration(self, batch_data):
        """One optimisation step (trainer.py:305-323)."""
        if getattr(self.model, "freeze", False):
            self._unhook_frozen_batchnorm()
        if isinstance(batch_data, _StaticBatch):
            if self._graph is not None and not self._graph.alive():
                self.drop_graph()
            if self._graph is None:
                self._graph = _StepGraph(self)
            self.current_loss = self._graph.step()
            return
        self.current_loss = loss = self.loss(batch_data)
        self.optimizer.zero_grad()
        seed = self.__dict__.get("_unit")
        if seed is None or seed.device != loss.device:
            seed = self.__dict__["_unit"] = torch.ones((), device=loss.device)
        loss.backward(gradient=seed)  # a persistent seed: autograd does not launch a fill for it
        self.optimizer.step()         # clip_grad_value_ (trainer.py:320-321) happens inside the fused kernel

    @staticmethod
    def _epoch_mean(values):
        if len(values) and torch.is_tensor(values[0]):
            return float(torch.stack(values).double().mean().item())
        return float(np.array(values).mean())

    # (log key, position in the model's loss vector) of what a captured step sums on the device; the subclass that
    # defines `loss` says which entries it logs
    _LOGGED = ()

    def _bank_graph_sums(self):
        """Move the captured steps' device-side loss sums into `_banked` (key -> [sum, steps])."""
        sums, n = self._graph.drain() if self._graph is not None else (None, 0)
        if n:
            bank = self.__dict__.setdefault("_banked", {})
            for key, pos in self._LOGGED:
                if key != "mmd_loss" or self.model.use_mmd:
                    entry = bank.setdefault(key, [0.0, 0])
                    entry[0] += sums[pos]
                    entry[1] += n

    def _fold_iter_logs(self, prefix):
        bank = {}
        if prefix == "epoch_":
            self._bank_graph_sums()
            bank = self.__dict__.pop("_banked", {})
        for key in list(bank) + [k for k in self.iter_logs if k not in bank]:
            values = self.iter_logs.get(key, [])
            if key in bank:  # steps replayed from the graph, plus whatever ran the ordinary way (a short last batch)
                total, n = bank[key]
                if len(values):
                    total += self._epoch_mean(values) * len(values)
                self.logs[prefix + key].append(total / (n + len(values)))
            else:
                self.logs[prefix + key].append(self._epoch_mean(values))

    def on_epoch_end(self):
        self._fold_iter_logs("epoch_")
        self.model._engine().check_status()
        if self.valid_data is not None:
            self.validate()
        if self.monitor:
            print_progress(self.epoch, self.logs, self.n_epochs, self.monitor_only_val)

    @torch.no_grad()
    def validate(self):
        """trainer.py:338-353: eval mode (dropout off, running BatchNorm statistics), z still sampled."""
        self.model.eval()
        self.iter_logs = defaultdict(list)
        for batch in self.dataloader_valid:
            self.loss(batch)
        self._fold_iter_logs("val_")
        self.model.train()

    def check_early_stop(self):
        """trainer.py:355-370: best-state bookkeeping, stop decision, learning-rate decay on a plateau."""
        monitor = self.early_stopping
        value = self.logs[monitor.early_stopping_metric][-1]
        if monitor.update_state(value):
            self.best_state_dict, self.best_epoch = self.model.state_dict(), self.epoch
        go_on, decay = monitor.step(value)
        if decay:
            print(f"\nLearning rate x {monitor.lr_factor}")
            for group in self.optimizer.param_groups:
                group["lr"] *= monitor.lr_factor
        return go_on


class Celligner2Trainer(Trainer):
    # (log key, position in forward_total's vector [recon, kl, mmd, ce, total, recon+kl+mmd]), in the order the
    # reference appends them (semisupervised.py:67-73)
    _LOGGED = (("loss", 4), ("unweighted_loss", 5), ("recon_loss", 0), ("kl_loss", 1), ("class_ce_loss", 3),
               ("mmd_loss", 2))

    def __init__(self, model, adata, **kwargs):
        super().__init__(model, adata, **kwargs)
        if self.graph and type(self).loss is not Celligner2Trainer.loss:
            raise ValueError("graph=True captures Celligner2Trainer's own loss; a subclass that overrides `loss` "
                             "must run with graph=False")

    def loss(self, total_batch=None):
        """semisupervised.py:63-74: `loss = recon + alpha*kl + mmd + ce` (:66) and `unweighted_loss = recon + kl + mmd`
        (:69), built together with the four losses by one kernel (Celligner2.forward_total) instead of single-element
        torch ops; the six log entries are device scalars (views, no kernel) unless log_sync == "step"."""
        total, vec = self.model.forward_total((1.0, self.calc_alpha_coeff(), 1.0, 1.0), **total_batch)
        if self.log_sync == "step":
            values = vec.tolist()                # one sync, python floats: the reference's six .item() calls
            self.model._engine().check_status()  # ... and its error timing: a bad batch raises here, not at epoch end
        else:
            values = vec.detach().unbind(0)
        for name, pos in self._LOGGED:
            if name != "mmd_loss" or self.model.use_mmd:
                self.iter_logs[name].append(values[pos])
        return total
