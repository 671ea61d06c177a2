"""Drop-in `Trainer` / `Celligner2Trainer` (mirror of celligner2/trainers/celligner2/trainer.py:13-370 and
semisupervised.py:4-74): same constructor kwargs, `train(n_epochs, lr, eps)`, overridable hooks, `logs` keys.

What differs underneath: the dataset is resident in HBM and batches are gathered by index on the device; the
step (loss -> zero_grad -> backward -> clip -> Adam) runs in the HIP library; per-iteration loss scalars stay on
the device and are averaged once per epoch (log_sync="epoch") instead of six .item() syncs per step.
"""
from __future__ import annotations

import sys
import time
from collections import defaultdict

import numpy as np
import torch

from .dataset import make_dataset
from .optim import FusedAdam


class EarlyStopping(object):
    """celligner2/utils/monitor.py:4-115 (host scalar logic, restated)."""

    def __init__(self, early_stopping_metric="val_unweighted_loss", mode="min", threshold=0, patience=20,
                 reduce_lr=True, lr_patience=13, lr_factor=0.1):
        self.early_stopping_metric = early_stopping_metric
        self.mode = mode
        self.threshold = threshold
        self.patience = patience
        self.reduce_lr = reduce_lr
        self.lr_patience = lr_patience
        self.lr_factor = lr_factor
        self.epoch = 0
        self.wait = 0
        self.wait_lr = 0
        self.current_performance = np.inf
        sign = 1 if mode == "min" else -1
        self.best_performance = sign * np.inf
        self.best_performance_state = sign * np.inf
        if patience == 0:
            self.is_better = lambda a, b: True
            self.step = lambda a: False

    def step(self, scalar):
        self.epoch += 1
        if self.epoch < self.patience:
            continue_training, lr_update = True, False
        elif self.wait >= self.patience:
            continue_training, lr_update = False, False
        else:
            if not self.reduce_lr:
                lr_update = False
            elif self.wait_lr >= self.lr_patience:
                lr_update = True
                self.wait_lr = 0
            else:
                lr_update = False
            self.current_performance = scalar
            if self.mode == "min":
                improvement = self.best_performance - self.current_performance
            else:
                improvement = self.current_performance - self.best_performance
            if improvement > 0:
                self.best_performance = self.current_performance
            if improvement < self.threshold:
                self.wait += 1
                self.wait_lr += 1
            else:
                self.wait = 0
                self.wait_lr = 0
            continue_training = True
        if not continue_training:
            print("\nStopping early: no improvement of more than " + str(self.threshold) + " nats in "
                  + str(self.patience) + " epochs")
        return continue_training, lr_update

    def update_state(self, scalar):
        if self.mode == "min":
            improved = (self.best_performance_state - scalar) > 0
        else:
            improved = (scalar - self.best_performance_state) > 0
        if improved:
            self.best_performance_state = scalar
        return improved


def print_progress(epoch, logs, n_epochs=10000, only_val_losses=True):
    """trainers/celligner2/_utils.py:13-76 (same text layout)."""
    message = ""
    for key in logs:
        if "unweighted" in key:
            continue
        if only_val_losses and "val_" not in key:
            continue
        message += f" - {key:s}: {logs[key][-1]:7.10f}"
    iteration, total, length = epoch + 1, n_epochs, 20
    percent = "{0:.1f}".format(100 * (iteration / float(total)))
    filled = int(length * iteration // total)
    bar = "█" * filled + "-" * (length - filled)
    sys.stdout.write("\r%s |%s| %s%s %s" % ("", bar, percent, "%", message))
    if iteration == total:
        sys.stdout.write("\n")
    sys.stdout.flush()


class _Loader:
    """Index stream of the reference's DataLoader(+WeightedRandomSampler / shuffle) served from a DeviceDataset.

    rng="reference" consumes torch's CPU generator exactly as torch.utils.data does (one int64 `random_()` per
    iterator, then `multinomial` / a re-seeded `randperm`), so a seeded run visits the same rows as the reference;
    rng="device" draws on the GPU generator."""

    def __init__(self, dataset, batch_size, weights=None, num_samples=None, rng="reference", xin_dtype=None,
                 apply_log=False, world=None, nb_const=False, resident_xin=None):
        self.nb_const = nb_const  # deliver the data-only NB term (sum goodloc*lgamma(x+1)) with every batch
        # callable(batch rows) -> bool: leave the encoder input in the dataset's resident bf16 matrix (read in place by the
        # row-indexed first-layer GEMMs) instead of writing a per-batch copy
        self.resident_xin = resident_xin or (lambda n: False)
        self.ds, self.bs, self.weights, self.num_samples, self.rng = dataset, batch_size, weights, num_samples, rng
        self.xin_dtype, self.apply_log = xin_dtype, apply_log
        self.world = world  # data-parallel: batch_size is the GLOBAL batch, each rank gathers its own slice
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
                # equal shares on every rank (the MMD all-gather needs them): a short last batch loses
                # at most world_size-1 rows
                per = sub.numel() // self.world.world_size
                if per == 0:
                    continue
                sub = sub[self.world.rank * per:(self.world.rank + 1) * per]
            yield self.ds.gather(sub, self.xin_dtype, self.apply_log, self._bufs.setdefault(sub.numel(), {}),
                                 self.nb_const, resident_rows=self.nb_const, resident_xin=self.resident_xin(sub.numel()))


class Trainer:
    def __init__(self, model, adata, condition_keys: list = None, predictor_keys: list = None,
                 cell_type_key: str = None, batch_size: int = 128, alpha_epoch_anneal: int = None,
                 alpha_kl: float = 0.001, use_early_stopping: bool = True, reload_best: bool = True,
                 early_stopping_kwargs: dict = None, min_weight: float = 0.2, use_leiden: bool = True,
                 leiden_neighbors: int = 8, leiden_resolution: float = 1, dataset_key="dataset", **kwargs):
        self.adata = adata
        self.model = model
        self.condition_keys = condition_keys
        self.predictor_keys = predictor_keys
        if cell_type_key is not None or not use_leiden:
            self.cell_type_key = cell_type_key
        else:
            raise NotImplementedError(
                "use_leiden=True needs scanpy's Leiden clustering (trainer.py:91-113), which is data preparation "
                "outside the accelerated path: pass cell_type_key=<obs column> or use_leiden=False")
        self.batch_size = batch_size
        self.alpha_epoch_anneal = alpha_epoch_anneal
        self.alpha_iter_anneal = kwargs.pop("alpha_iter_anneal", None)
        self.use_early_stopping = use_early_stopping
        self.reload_best = reload_best
        self.alpha_kl = alpha_kl
        early_stopping_kwargs = early_stopping_kwargs if early_stopping_kwargs else dict()
        self.n_samples = kwargs.pop("n_samples", None)
        self.train_frac = kwargs.pop("train_frac", 0.7)
        self.use_stratified_sampling = kwargs.pop("use_stratified_sampling", True)
        self.weight_decay = kwargs.pop("weight_decay", 0.04)
        self.clip_value = kwargs.pop("clip_value", 0.0)
        self.n_workers = kwargs.pop("n_workers", 0)  # accepted, unused: batches are gathered on the device
        self.seed = kwargs.pop("seed", 2020)
        self.monitor = kwargs.pop("monitor", True)
        self.monitor_only_val = kwargs.pop("monitor_only_val", True)
        # build-only knobs (reference-preserving defaults)
        self.log_sync = kwargs.pop("log_sync", "epoch")  # "step": .item() every iteration like the reference
        self.rng = kwargs.pop("rng", "reference")        # index stream / noise on torch's CPU generator
        self.early_stopping = EarlyStopping(**early_stopping_kwargs)
        torch.manual_seed(self.seed)
        if not torch.cuda.is_available():
            raise RuntimeError("celligner2_amd needs an MI355X GPU: there is no CPU training path")
        torch.cuda.manual_seed(self.seed)
        self.device = torch.device("cuda", torch.cuda.current_device())
        self.model.to(self.device)
        self.model.rng_mode = self.rng
        self.min_weight = min_weight
        self.epoch = -1
        self.n_epochs = None
        self.iter = 0
        self.best_epoch = None
        self.best_state_dict = None
        self.current_loss = None
        self.previous_loss_was_nan = False
        self.nan_counter = 0
        self.optimizer = None
        self.training_time = 0
        self.sampler = None
        self.dataloader_train = None
        self.dataloader_valid = None
        self.iters_per_epoch = None
        self.val_iters_per_epoch = None
        self.logs = defaultdict(list)
        self.world = kwargs.pop("world", None)  # parallel.DataParallelStep (one process per GPU) or None
        if self.world is not None:
            self.world.attach_model(self.model)
        if "train_data" in kwargs:  # DeviceDataset objects staged by the caller (adata may be None)
            self.train_data, self.valid_data = kwargs.pop("train_data"), kwargs.pop("valid_data", None)
            return
        self.train_data, self.valid_data = make_dataset(
            self.adata, train_frac=self.train_frac, condition_keys=self.condition_keys,
            cell_type_key=self.cell_type_key, condition_encoder=self.model.condition_encoder,
            cell_type_encoder=self.model.cell_type_encoder, predictor_keys=self.predictor_keys,
            predictor_encoder=self.model.predictor_encoder, min_weight=self.min_weight, device=self.device)

    # ------------------------------------------------------------------ loaders (trainer.py:184-229)
    def initialize_loaders(self):
        if self.n_samples is None or self.n_samples > len(self.train_data):
            self.n_samples = len(self.train_data)
        self.iters_per_epoch = int(np.ceil(self.n_samples / self.batch_size))
        eng = self.model._engine()
        xin_dtype = eng.enc[0].gt
        log_in = self.model.applylog and self.model.recon_loss != "mse"
        if not log_in and xin_dtype == torch.float32:
            xin_dtype = None  # the encoder reads x itself
        nbc = self.model.recon_loss == "nb"
        G, H0 = self.model.input_dim, self.model.hidden_layer_sizes[0]

        def rx(n):  # both first-layer GEMMs (forward [n,H0] over G, weight gradient [H0,G] over n) on the 256x256 kernel
            from . import ops
            return (nbc and xin_dtype == torch.bfloat16 and n % 8 == 0
                    and ops.gemm_tile(torch.bfloat16, n, H0, G) == 256 and ops.gemm_tile(torch.bfloat16, H0, G, n) == 256)

        if self.use_stratified_sampling:
            weights = torch.tensor(self.train_data.stratifier_weights)
            self.dataloader_train = _Loader(self.train_data, self.batch_size, weights, self.n_samples, self.rng,
                                            xin_dtype, log_in, self.world, nbc, rx)
        else:
            self.dataloader_train = _Loader(self.train_data, self.batch_size, None, None, self.rng, xin_dtype, log_in,
                                            self.world, nbc, rx)
        if self.valid_data is not None:
            val_bs = min(self.batch_size, len(self.valid_data))
            self.val_iters_per_epoch = int(np.ceil(len(self.valid_data) / self.batch_size))
            self.dataloader_valid = _Loader(self.valid_data, val_bs, None, None, self.rng, xin_dtype, log_in, self.world,
                                            nbc, rx)

    def make_optimizer(self, lr=1e-3, eps=0.01):
        """trainer.py:262-266: Adam over the parameters that require a gradient, L2 weight decay; here the fused
        multi-tensor kernel, wired to the engine's operand copies and (data-parallel runs) to the gradient exchange."""
        lookup, updated = self.model._adam_hooks()
        return FusedAdam([p for p in self.model.parameters() if p.requires_grad], lr=lr, eps=eps,
                         weight_decay=self.weight_decay, clip_value=self.clip_value, shadow_lookup=lookup,
                         on_updated=updated, grad_sync=self.world)

    def calc_alpha_coeff(self):
        """trainer.py:231-253."""
        if self.alpha_epoch_anneal is not None:
            return min(self.alpha_kl * self.epoch / self.alpha_epoch_anneal, self.alpha_kl)
        if self.alpha_iter_anneal is not None:
            return min(self.alpha_kl * (self.epoch * self.iters_per_epoch + self.iter) / self.alpha_iter_anneal,
                       self.alpha_kl)
        return self.alpha_kl

    def train(self, n_epochs: int = 400, lr: float = 1e-3, eps: float = 0.01):
        self.initialize_loaders()
        begin = time.time()
        self.model.train()
        self.n_epochs = n_epochs
        eng = self.model._engine()
        self.optimizer = self.make_optimizer(lr, eps)
        self.before_loop()
        for self.epoch in range(n_epochs):
            self.on_epoch_begin(lr, eps)
            self.iter_logs = defaultdict(list)
            for self.iter, batch_data in enumerate(self.dataloader_train):
                self.on_iteration(batch_data)
            self.on_epoch_end()
            if self.use_early_stopping:
                if not self.check_early_stop():
                    break
        if self.best_state_dict is not None and self.reload_best:
            print("Saving best state of network...")
            print("Best State was in Epoch", self.best_epoch)
            self.model.load_state_dict(self.best_state_dict)
        self.model.eval()
        self.after_loop()
        eng.check_status()
        self.training_time += time.time() - begin

    def before_loop(self):
        pass

    def on_epoch_begin(self, lr, eps):
        pass

    def after_loop(self):
        pass

    def on_iteration(self, batch_data):
        """trainer.py:305-323."""
        if getattr(self.model, "freeze", False):  # :307-312: BatchNorm layers frozen by surgery stop tracking
            for module in self.model.modules():
                if isinstance(module, torch.nn.BatchNorm1d) and not module.weight.requires_grad:
                    module.affine = False
                    module.track_running_stats = False
        self.current_loss = loss = self.loss(batch_data)
        self.optimizer.zero_grad()
        one = self.__dict__.get("_one")
        if one is None or one.device != loss.device:
            one = self.__dict__["_one"] = torch.ones((), device=loss.device)
        loss.backward(gradient=one)  # a persistent seed: autograd does not launch a fill for it
        # clip_grad_value_ (trainer.py:320-321) is folded into the fused Adam kernel (clip_value hyper-parameter)
        self.optimizer.step()

    def _mean_log(self, values):
        if len(values) and torch.is_tensor(values[0]):
            return float(torch.stack(values).double().mean().item())
        return float(np.array(values).mean())

    def on_epoch_end(self):
        for key in self.iter_logs:
            self.logs["epoch_" + key].append(self._mean_log(self.iter_logs[key]))
        self.model._engine().check_status()
        if self.valid_data is not None:
            self.validate()
        if self.monitor:
            print_progress(self.epoch, self.logs, self.n_epochs, self.monitor_only_val)

    @torch.no_grad()
    def validate(self):
        self.model.eval()
        self.iter_logs = defaultdict(list)
        for batch_data in self.dataloader_valid:
            _ = self.loss(batch_data)
        for key in self.iter_logs:
            self.logs["val_" + key].append(self._mean_log(self.iter_logs[key]))
        self.model.train()

    def check_early_stop(self):
        metric = self.early_stopping.early_stopping_metric
        if self.early_stopping.update_state(self.logs[metric][-1]):
            self.best_state_dict = self.model.state_dict()  # alias, not a copy: as the reference (trainer.py:359)
            self.best_epoch = self.epoch
        continue_training, update_lr = self.early_stopping.step(self.logs[metric][-1])
        if update_lr:
            print("\nADJUSTED LR")
            for param_group in self.optimizer.param_groups:
                param_group["lr"] *= self.early_stopping.lr_factor
        return continue_training


class Celligner2Trainer(Trainer):
    def __init__(self, model, adata, **kwargs):
        super().__init__(model, adata, **kwargs)

    def loss(self, total_batch=None):
        """semisupervised.py:63-74."""
        # loss = recon + alpha*kl + mmd + ce (:66) and unweighted_loss = recon + kl + mmd (:69), built with the four
        # losses by one kernel (Celligner2.forward_total) instead of single-element torch ops
        loss, v = self.model.forward_total((1.0, self.calc_alpha_coeff(), 1.0, 1.0), **total_batch)
        if self.log_sync == "step":
            vals = v.tolist()
        else:
            vals = list(v.detach().unbind(0))  # views: no kernel
        r, k, m, c = vals[0], vals[1], vals[2], vals[3]
        self.iter_logs["loss"].append(vals[4])
        self.iter_logs["unweighted_loss"].append(vals[5])
        self.iter_logs["recon_loss"].append(r)
        self.iter_logs["kl_loss"].append(k)
        self.iter_logs["class_ce_loss"].append(c)
        if self.model.use_mmd:
            self.iter_logs["mmd_loss"].append(m)
        return loss
