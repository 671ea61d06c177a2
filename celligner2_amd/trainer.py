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
"""
from __future__ import annotations

import sys
import time
from collections import defaultdict

import numpy as np
import torch

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


class _Loader:
    """Index stream of the reference's DataLoader(+WeightedRandomSampler / shuffle) served from a DeviceDataset.

    rng="reference" consumes torch's CPU generator exactly as torch.utils.data does (one int64 `random_()` per
    iterator, then `multinomial` / a re-seeded `randperm`), so a seeded run visits the same rows as the reference;
    rng="device" draws on the GPU generator.  Like the reference's loader the last batch of an epoch may be short.

    Data-parallel runs (`world`): `batch_size` is the GLOBAL batch; every rank draws the same index stream and keeps
    its own contiguous share of each global batch.  Shares are equal (the MMD all-gather needs that), so a short last
    batch loses its last `len % world_size` rows and a last batch with fewer rows than ranks is skipped."""

    def __init__(self, dataset, batch_size, weights=None, num_samples=None, rng="reference", xin_dtype=None,
                 apply_log=False, world=None, nb_const=False, resident_xin=None):
        self.nb_const = nb_const  # deliver the data-only NB term (sum goodloc*lgamma(x+1)) with every batch
        # callable(batch rows) -> bool: leave the encoder input in the dataset's resident bf16 matrix (read in place by the
        # row-indexed first-layer GEMMs) instead of writing a per-batch copy
        self.resident_xin = resident_xin or (lambda n: False)
        self.ds, self.bs, self.weights, self.num_samples, self.rng = dataset, batch_size, weights, num_samples, rng
        self.xin_dtype, self.apply_log = xin_dtype, apply_log
        self.world = world
        self._bufs = {}

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
            return torch.multinomial(self.weights.to(dev), self.num_samples, True)
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
            yield self.ds.gather(sub, self.xin_dtype, self.apply_log, self._bufs.setdefault(sub.numel(), {}),
                                 self.nb_const, resident_rows=self.nb_const, resident_xin=self.resident_xin(sub.numel()))


# keyword -> default of the options the reference pops from **kwargs (trainer.py:117-137), then the build-only ones
_POPPED = (("alpha_iter_anneal", None), ("n_samples", None), ("train_frac", 0.7), ("use_stratified_sampling", True),
           ("weight_decay", 0.04), ("clip_value", 0.0), ("n_workers", 0), ("seed", 2020), ("monitor", True),
           ("monitor_only_val", True),
           ("log_sync", "epoch"),   # "step": python floats every iteration (and device errors raised at once)
           ("rng", "reference"),    # index stream / noise on torch's CPU generator, in the reference's order
           ("world", None),         # parallel.DataParallelStep of a one-process-per-GPU run
           ("graph", False))        # replay the step as a captured HIP graph (fixed-shape batches only)


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
        xin_dtype = model._engine().enc[0].gt
        log_in = model.applylog and model.recon_loss != "mse"
        if not log_in and xin_dtype == torch.float32:
            xin_dtype = None  # the encoder reads x itself
        nb = model.recon_loss == "nb"
        G, H0 = model.input_dim, model.hidden_layer_sizes[0]

        def in_place(n):  # both first-layer GEMMs (forward [n,H0] over G, weight gradient [H0,G] over n) on the 256x256 kernel
            from . import ops
            return (nb and xin_dtype == torch.bfloat16 and n % 8 == 0
                    and ops.gemm_tile(torch.bfloat16, n, H0, G) == 256 and ops.gemm_tile(torch.bfloat16, H0, G, n) == 256)

        return _Loader(data, batch_size, weights, n_samples, self.rng, xin_dtype, log_in, self.world, nb, in_place)

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
        if self.valid_data is not None:
            self.val_iters_per_epoch = int(np.ceil(len(self.valid_data) / self.batch_size))
            self.dataloader_valid = self._loader(self.valid_data, min(self.batch_size, len(self.valid_data)), None, None)

    def make_optimizer(self, lr=1e-3, eps=0.01):
        """trainer.py:262-266: Adam over the parameters that require a gradient, L2 weight decay; here the fused
        multi-tensor kernel, wired to the engine's operand copies and (data-parallel runs) to the gradient exchange."""
        lookup, updated = self.model._adam_hooks()
        return FusedAdam([p for p in self.model.parameters() if p.requires_grad], lr=lr, eps=eps,
                         weight_decay=self.weight_decay, clip_value=self.clip_value, shadow_lookup=lookup,
                         on_updated=updated, grad_sync=self.world)

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

    def _fold_iter_logs(self, prefix):
        for key, values in self.iter_logs.items():
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
