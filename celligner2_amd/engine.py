"""Host-side driver of one training step: orders the C-ABI kernels of libc2mi.so.

`StepEngine` owns no parameters: it is handed the module's parameter tensors (by state_dict
name) and keeps only workspaces, bf16 weight shadows and gradient buffers.  It restates, call
by call, the reference's `Celligner2.forward` (celligner2/model/celligner2.py:259-374) and the
autograd graph torch builds for it; every FLOP and byte is moved by the HIP kernels.

Precision modes
  fp32: operands of every GEMM are fp32 (v_mfma_f32_32x32x2_f32, exact fp32 products)
  bf16: GEMM operands (activations, weight shadows, back-propagated gradients) are bf16,
        accumulation / losses / parameter gradients / Adam state stay fp32.
"""
from __future__ import annotations

import contextlib
from collections import OrderedDict

import torch

from . import ops

F32 = torch.float32


class _Block:
    """Linear (+bias) (+cond) -> LayerNorm -> ReLU -> Dropout (modules.py:77-107,124-130,298-312)."""

    def __init__(self, name, w, b, wc, drop_name, use_cond, norm=None):
        self.name = name  # e.g. "enc0"
        self.w, self.b, self.wc = w, b, wc  # parameter NAMES (b / wc may be None)
        self.norm = norm  # use_bn: module name of the block's BatchNorm1d (its .weight / .bias are parameters)
        self.drop_name = drop_name
        self.use_cond = use_cond
        self.gt = F32     # operand dtype of this block's own GEMMs (fwd input, dh, weight)
        self.out_t = F32  # dtype of its activation = operand dtype of the consumer GEMM


class _T:
    """A [B, H] operand held as its transposed copy t = [H, B] (StepEngine._narrow_t)."""

    def __init__(self, t):
        self.t = t

    @property
    def shape(self):
        return (self.t.shape[1], self.t.shape[0])

    @property
    def dtype(self):
        return self.t.dtype

    def cols(self, sl):
        return _T(self.t[sl])


def _cols(x, sl):
    return x.cols(sl) if isinstance(x, _T) else x[:, sl]


class StepEngine:
    def __init__(self, cfg, params, precision="fp32", bn_modules=None):
        """cfg: dict of model options (see model.Celligner2._cfg); params: {name: Parameter};
        bn_modules: {module name: BatchNorm1d holder} when cfg["use_bn"]."""
        assert precision in ("fp32", "bf16")
        self.cfg = cfg
        self.p = params
        self.precision = precision
        self.T = F32 if precision == "fp32" else torch.bfloat16
        self.x3 = precision == "bf16"  # narrow fp32-operand GEMMs run as 3 bf16 MFMA products in bf16 mode
        self.dev = next(iter(params.values())).device
        self._ws = {}
        self._pool = {}       # key -> OrderedDict[(shape, dtype)] -> tensor, least recently used first (see buf)
        self.recorded = None  # list while a step is being captured: every scratch tensor it uses
        self._shadow = {}
        self._shadow_ver = {}
        self._wplanes = {}   # bf16x3: (version, [2,N,K] bf16 planes) of the fp32 weights of the narrow layers
        self._aplanes = {}   # bf16x3: planes of this step's fp32 activations / gradients, by (data_ptr, shape)
        self.grad = {}
        self._slab_jobs = None  # list while a backward pass runs: (slabs, gradient view) of split weight gradients to be summed
        self._touched = set()
        self.world_size, self.rank = 1, 0
        self.comm = None  # set by parallel.DataParallelStep
        self.exchange = False  # collectives are issued: more than one rank (or the single-rank RCCL rehearsal)
        self.bn = dict(bn_modules or {})
        self.use_bn = bool(cfg.get("use_bn"))
        nh = len(cfg["hidden"])
        C = cfg["n_conditions"]
        nm = (lambda s: s) if self.use_bn else (lambda s: None)  # noqa: E731
        self.enc = []
        for i in range(nh):
            if i == 0:
                self.enc.append(_Block("enc0", "encoder.FC.L0.expr_L.weight", "encoder.FC.L0.expr_L.bias",
                                       "encoder.FC.L0.cond_L.weight" if C else None, "enc.D0", bool(C),
                                       nm("encoder.FC.N0")))
            else:
                self.enc.append(_Block(f"enc{i}", f"encoder.FC.L{i}.weight", f"encoder.FC.L{i}.bias", None,
                                       f"enc.D{i}", False, nm(f"encoder.FC.N{i}")))
        self.dec = [_Block("dec0", "decoder.FirstL.L0.expr_L.weight", None,
                           "decoder.FirstL.L0.cond_L.weight" if C else None, "dec.D0", bool(C),
                           nm("decoder.FirstL.N0"))]
        for i in range(1, nh):
            self.dec.append(_Block(f"dec{i}", f"decoder.HiddenL.L{i}.weight", None, None, f"dec.D{i}", False,
                                   nm(f"decoder.HiddenL.N{i}")))
        self.cls = [_Block("cls0", "classifier.FirstL.L0.weight", None, None, "cls.D0", False,
                           nm("classifier.FirstL.N0"))]
        for i in range(1, len(cfg["cls_hidden"])):
            self.cls.append(_Block(f"cls{i}", f"classifier.HiddenL.L{i}.weight", None, None, f"cls.D{i}", False,
                                   nm(f"classifier.HiddenL.N{i}")))
        for chain in (self.enc, self.dec, self.cls):
            for blk in chain:
                if blk.norm and blk.norm not in self.bn:
                    raise ValueError(f"use_bn: no BatchNorm module given for {blk.norm}")
        self.head = "decoder.recon_decoder.0" if cfg["recon_loss"] == "mse" else "decoder.mean_decoder.0"
        self.head2 = "decoder.dropout_decoder" if cfg["recon_loss"] == "zinb" else None  # ZINB dropout logits
        # Mixed-precision policy of bf16 mode: only the gene-width GEMMs (>= WIDE on either weight dimension:
        # encoder L0 and the decoder output layer, 96.6 % of the FLOPs at 19k genes) take bf16 operands; the
        # narrow hidden GEMMs keep fp32 operands, so the back-propagated signal is rounded to bf16 only where it
        # meets a gene-width contraction.
        self.head_t = self.gemm_dtype(self.head + ".weight")
        chains = (self.enc, self.dec, self.cls)
        for chain in chains:
            for blk in chain:
                blk.gt = self.gemm_dtype(blk.w)
            for a, b in zip(chain[:-1], chain[1:]):
                a.out_t = b.gt
        # bf16 mode: a gene-width first layer reads its input CENTRED (minus a per-gene vector c), the centre's share W c
        # goes into the bias in fp32 and (sum_b dh_b) (x) c into the weight gradient -- the same layer, but the operand that
        # is rounded to bf16 no longer carries the per-gene level all cells share (include/c2mi.h, c2_col_mean)
        import os
        off = lambda name: os.environ.get(name) == "0"  # noqa: E731  (A/B switches: measurements and tests)
        self.center_input = self.enc[0].gt == torch.bfloat16 and not off("C2_CENTER")
        self._center = None
        # ... and, when that input is log(1 + x) (bounded by 89: in IEEE half's range whatever the counts), the first layer's
        # FORWARD reads input and weight as IEEE halves (C2_F16: 11 significant bits instead of bf16's 8, the same matrix
        # rate).  Measured at 19 000 genes against the fp64 CPU restatement (tests/test_benchwidth_*_gpu.py): the 2^-9 rounding
        # of a bf16 encoder input put the latent heads' gradients at 1.8-3 %, beyond bf16 mode's 1e-2.  The layer's weight
        # gradient keeps bf16 operands (dh is a gradient: range matters more than digits), so the input exists in both forms.
        log_in = bool(cfg["applylog"]) and cfg["recon_loss"] != "mse"
        self.fwd16 = torch.float16 if (self.enc[0].gt == torch.bfloat16 and log_in and not off("C2_FWD16")) else None
        self.half_logits_on = not off("C2_HALF_LOGITS")
        self.rows4_on = not off("C2_ROWS4")  # padded-row route of the NB / ZINB likelihood at gene counts that are not multiples of 4
        # The backward pass is two chains: activation gradients (tail, input-gradient GEMM, the next layer's tail ...: each
        # waits for the one before) and parameter gradients (bias / condition column sums, weight-gradient GEMM, slab sums:
        # nothing but the optimizer waits for them).  The second chain is queued on a stream of its own (one process, no
        # gradient exchange), so its launches fill the compute units the narrow layers of the first chain leave idle; a
        # captured step keeps the two branches.  C2_SIDE=0: one stream.
        self.side_on = os.environ.get("C2_SIDE") == "1"
        self.side_level = int(os.environ.get("C2_SIDE_LEVEL", "1"))
        self._wstream = None
        self._side_used = False
        self.enc[-1].out_t = self.gemm_dtype("encoder.mean_encoder.weight")  # feeds the mean / log-var heads
        self.dec[-1].out_t = self.head_t
        self.cls[-1].out_t = self.gemm_dtype("classifier.classifier.0.weight")

    WIDE = 2048

    def input_spec(self):
        """Keywords of DeviceDataset.gather that deliver the encoder input in the form this engine reads: operand dtype of
        the first layer's forward (None: it reads x itself), centred or not, with or without the bf16 twin."""
        dt = self.fwd16 or self.enc[0].gt
        log_in = bool(self.cfg["applylog"]) and self.cfg["recon_loss"] != "mse"
        if dt == F32 and not log_in:
            dt = None
        return dict(xin_dtype=dt, centered=self.center_input, xin2=self.fwd16 is not None)

    def gemm_dtype(self, wname):
        if self.T == F32:
            return F32
        return self.T if max(self.p[wname].shape) >= self.WIDE else F32

    # ------------------------------------------------------------------ buffers
    KEEP_SHAPES = 4  # train / validation batches, each full-size and short

    def buf(self, key, shape, dtype=F32, zero=False):
        """The scratch tensor `key` in this shape: self._ws[key] afterwards.  Every key keeps its KEEP_SHAPES most
        recently used shapes, so alternating batch sizes (the short last batch of an epoch, validation) do not free and
        re-allocate the workspace twice per epoch -- and a step captured as a HIP graph for one batch size keeps its
        addresses while other sizes run.  While `self.recorded` is a list, every tensor handed out is appended to it (a
        captured graph keeps that list: what it wrote to stays allocated even if the pool later lets go of it)."""
        shapes = self._pool.get(key)
        if shapes is None:
            shapes = self._pool[key] = OrderedDict()
        slot = (tuple(int(s) for s in shape), dtype)
        t = shapes.get(slot)
        if t is None:
            t = shapes[slot] = (torch.zeros if zero else torch.empty)(slot[0], dtype=dtype, device=self.dev)
            if len(shapes) > self.KEEP_SHAPES:
                shapes.popitem(last=False)
        else:
            shapes.move_to_end(slot)
        self._ws[key] = t
        if self.recorded is not None:
            self.recorded.append(t)
        return t

    def abuf(self, key, shape, dtype):
        """A [B, H] activation that a gene-width GEMM may read as a 2-byte operand: rows ops.pitch2(H) elements apart (aligned
        rows, whole K-tiles: see there), the pad columns zeroed once, when the buffer is made, and never written again (the
        producers write H columns), so a reduction over the padded width adds exact zeros.  Returns the [B, H] view."""
        B, H = shape
        ld = ops.pitch2(H)
        if dtype == F32 or ld == H:
            return self.buf(key, shape, dtype)
        t = self.buf(key, (B, ld), dtype, zero=True)
        v = self._ws[key] = t[:, :H]
        return v

    @staticmethod
    def _padded_width(t):
        """Row stride of a 2-byte [rows, cols] view made by abuf / weight() / the dataset over a zero-padded buffer, else 0."""
        if t.dim() == 2 and t.element_size() == 2 and t.stride(1) == 1 and t.stride(0) != t.shape[1] and \
                t.stride(0) == ops.pitch2(t.shape[1]):
            return t.stride(0)
        return 0

    def _readiness_order(self):
        """Parameter names in the order the backward pass finishes their gradients."""
        order = [self.head + ".weight", self.head + ".bias", "theta"]
        if self.head2:
            order += [self.head2 + ".weight", self.head2 + ".bias"]
        gb = lambda blk: [blk.norm + ".weight", blk.norm + ".bias"] if blk.norm else []  # noqa: E731
        for blk in reversed(self.dec):
            order += gb(blk) + [blk.w, blk.wc]
        order += ["classifier.classifier.0.weight", "classifier.classifier.0.bias"]
        for blk in reversed(self.cls):
            order += gb(blk) + [blk.w]
        order += ["encoder.mean_encoder.weight", "encoder.log_var_encoder.weight", "encoder.mean_encoder.bias",
                  "encoder.log_var_encoder.bias"]  # (pairs adjacent: _build_arena glues them)
        for blk in reversed(self.enc):
            order += gb(blk) + [blk.w, blk.b, blk.wc]
        seen, out = set(), []
        for n in order:
            if n and n in self.p and n not in seen:
                seen.add(n)
                out.append(n)
        out += [n for n in self.p if n not in seen]
        return out

    def _build_arena(self):
        """One flat fp32 gradient arena (views per parameter) cut into three exchange buckets:
        [decoder output layer + theta] [everything between] [encoder L0].

        With a sharding communicator (parallel.DataParallelStep(shard_optimizer=True)) every bucket's length is a
        multiple of 64 x world_size, so that it cuts into equal 256-byte-aligned chunks, one per rank, and the trainable
        parameters themselves become views of a second flat buffer with the SAME offsets (`param_flat`): the bucket is
        reduce-scattered, each rank runs Adam on its chunk, and the chunk is all-gathered in place."""
        sharded = self.exchange and getattr(self.comm, "sharded", False)
        key = (tuple((n, tuple(p.shape), p.requires_grad) for n, p in self.p.items()), sharded, self.world_size)
        if getattr(self, "_arena_key", None) == key:
            return
        self._arena_key = key
        names = [n for n in self._readiness_order() if self.p[n].requires_grad]
        first = {self.head + ".weight", self.head + ".bias", "theta"}
        if self.head2:
            first |= {self.head2 + ".weight", self.head2 + ".bias"}
        last = {self.enc[0].w, self.enc[0].b, self.enc[0].wc}
        if self.enc[0].norm:
            last |= {self.enc[0].norm + ".weight", self.enc[0].norm + ".bias"}
        quantum = 64 * (self.world_size if sharded else 1)
        up = lambda v: (v + quantum - 1) // quantum * quantum  # noqa: E731
        part = lambda n: 0 if n in first else (2 if n in last else 1)  # noqa: E731
        # the two latent heads' weights (and biases) lie back to back, so that one [2L, H] ([2L]) view spans both
        glued = {self.HEADS[1] + ".weight": self.HEADS[0] + ".weight", self.HEADS[1] + ".bias": self.HEADS[0] + ".bias"}
        offs, total, prev = {}, 0, None
        for n in names:
            if prev is not None and part(n) != part(prev):
                total = up(total)  # a bucket starts on a chunk boundary
            if prev is not None and glued.get(n) == prev and self.p[prev].numel() % 4 == 0:  # (16-byte alignment)
                total = offs[prev] + self.p[prev].numel()
            offs[n] = total
            total += (self.p[n].numel() + 63) // 64 * 64
            prev = n
        total = up(total)
        self.grad_flat = torch.zeros(max(total, 1), dtype=F32, device=self.dev)
        self.grad = {n: self.grad_flat[offs[n]:offs[n] + self.p[n].numel()].view(self.p[n].shape) for n in names}
        ends = {n: offs[n] + (self.p[n].numel() + 63) // 64 * 64 for n in names}
        b0 = up(max([ends[n] for n in names if n in first], default=0))        # readiness order: `first` leads,
        lo2 = min([offs[n] for n in names if n in last], default=total)        # `last` closes the arena
        self.buckets = [(0, b0), (b0, lo2), (lo2, total)]
        self.param_flat = None
        if sharded:
            flat = torch.zeros(max(total, 1), dtype=F32, device=self.dev)
            for n in names:
                prm = self.p[n]
                view = flat[offs[n]:offs[n] + prm.numel()].view(prm.shape)
                view.copy_(prm.detach())
                prm.data = view
            self.param_flat = flat
            self._ws.pop("heads.W", None)  # the stacked view of the heads is rebuilt over the new storage
            self._ws.pop("heads.b", None)
            self._wplanes.pop(self.HEADW, None)
            self.comm.set_arena(self.grad_flat, flat, self.buckets)

    def gbuf(self, name):
        self._build_arena()
        self._touched.add(name)
        return self.grad[name]

    def _reduce_bucket(self, i):
        self._flush_slab_sums()  # a bucket is complete only once the split weight gradients that wait are summed
        if self.exchange:
            lo, hi = self.buckets[i]
            if hi > lo:
                self.comm.reduce_bucket_async(self.grad_flat[lo:hi])

    def weight(self, name):
        """GEMM operand view of a weight: the fp32 parameter itself, or its bf16 shadow."""
        if name == self.HEADW:
            return self._ws["heads.W"]
        p = self.p[name]
        if self.gemm_dtype(name) == F32:
            return p.detach()
        wdt = self.fwd16 if (self.fwd16 is not None and name == self.enc[0].w) else torch.bfloat16
        sh = self._shadow.get(name)
        if sh is None or sh.shape != p.shape:
            rp, ldp = (ops.pitch2(p.shape[0]), ops.pitch2(p.shape[1])) if p.dim() == 2 else (0, 0)
            if p.dim() == 2 and (rp != p.shape[0] or ldp != p.shape[1]):
                # rows ops.pitch2(width) elements apart, pad zeroed once (see abuf), and zero ROWS up to pitch2(row count): the
                # output layer's [G, H] is reduced over its rows by the input gradient, together with dlogits' padded columns
                # (_dgrad).  The Adam kernel refreshes such a copy through its row pitch when the width is a multiple of 4
                # (shadow_for_adam), else it is converted here after every optimizer step.
                sh = torch.zeros((rp, ldp), dtype=wdt, device=self.dev)[:p.shape[0], :p.shape[1]]
            else:
                sh = torch.empty(p.shape, dtype=wdt, device=self.dev)
            self._shadow[name] = sh
            self._shadow_ver[name] = None
        if self._shadow_ver[name] != p._version:
            if sh.is_contiguous():
                ops.cast_bf16(p.detach(), sh)
            else:
                ops.prep_input(p.detach(), sh, False)
            self._shadow_ver[name] = p._version
        return sh

    def shadow_for_adam(self, name):
        """What the fused Adam kernel refreshes together with the fp32 master weight: the bf16 shadow of a gene-width
        layer, or the (hi, lo) bf16 planes of a narrow layer's weight (the kernel writes through raw pointers, so
        torch's version counter does not move: without this the operand copies would go stale after a step)."""
        if self.T == F32:
            return None
        sh = self._shadow.get(name)
        if sh is not None:
            # (a copy in padded storage: the kernel maps its flat index through the row pitch when the width is a multiple
            # of 4; else weight() converts again when the version has moved)
            if sh.is_contiguous() or (sh.dim() == 2 and sh.stride(0) % 4 == 0 and sh.stride(1) == 1):
                return sh
            return None
        # (planes in padded storage -- a width that is not a multiple of 8 -- are not dense: the kernel's flat indexing
        # does not apply, the engine splits such a weight again when its version has moved)
        c = self._wplanes.get(name)
        if c is not None:
            return self._adam_planes(c[1][0], c[1][1])
        if name in (self.HEADS[0] + ".weight", self.HEADS[1] + ".weight"):
            c = self._wplanes.get(self.HEADW)  # rows [0,L) / [L,2L) of the stacked planes: (hi, lo) of this head
            if c is not None:
                L = self.p[name].shape[0]
                k = 0 if name.startswith(self.HEADS[0]) else 1
                return self._adam_planes(c[1][0, k * L:(k + 1) * L], c[1][1, k * L:(k + 1) * L])
        return None

    @staticmethod
    def _adam_planes(hi, lo):
        """(hi, lo) as the fused Adam kernel takes them: dense planes, or rows a pitch apart when the width is a multiple of 4
        (planes in padded storage); None otherwise -- the engine then splits the weight again when its version has moved."""
        if hi.is_contiguous() and lo.is_contiguous():
            return (hi, lo)
        if hi.stride(0) % 4 == 0 and hi.stride(1) == 1 and lo.stride() == hi.stride():
            return (hi, lo)
        return None

    def copies_current(self, name):
        """The optimizer has just refreshed this parameter's operand copy in its own kernel and advanced the
        parameter's version counter: record that version, so that the next forward does not convert again."""
        p = self.p[name]
        if name in self._shadow:
            self._shadow_ver[name] = p._version
        c = self._wplanes.get(name)
        if c is not None:
            self._wplanes[name] = (p._version, c[1])
        c = self._wplanes.get(self.HEADW)
        if c is not None and name in (self.HEADS[0] + ".weight", self.HEADS[1] + ".weight"):
            self._wplanes[self.HEADW] = (self._heads_version(), c[1])

    def as_dtype(self, key, t, dtype):
        """Operand view of an fp32 activation in `dtype` (a bf16 copy when the consumer GEMM is gene-width)."""
        if t.dtype == dtype:
            return t
        out = self.buf(key, t.shape, dtype)
        ops.prep_input(t, out, False)
        return out

    # ------------------------------------------------------------------ fused latent heads
    # mean_encoder and log_var_encoder read the same activation and have the same shape: they run as ONE GEMM
    # [B, 2L] = henc @ [W_mean; W_logvar]^T (forward, dgrad, wgrad).  The two weight (bias) parameters are made
    # views of one [2L, H] ([2L]) buffer -- state_dict, optimizer and checkpoints see two ordinary parameters.
    HEADS = ("encoder.mean_encoder", "encoder.log_var_encoder")
    HEADW = "@heads"  # virtual weight name of the stacked matrix

    @staticmethod
    def _back_to_back(a, b):
        return (a.untyped_storage().data_ptr() == b.untyped_storage().data_ptr() and a.is_contiguous()
                and b.is_contiguous() and b.data_ptr() == a.data_ptr() + a.numel() * a.element_size())

    def _heads_fusable(self):
        wm, wl = (self.p[h + ".weight"] for h in self.HEADS)
        bm, bl = (self.p[h + ".bias"] for h in self.HEADS)
        ok = (wm.shape == wl.shape and self.gemm_dtype(self.HEADS[0] + ".weight") == F32
              and wm.requires_grad == wl.requires_grad and bm.requires_grad == bl.requires_grad
              and wm.dtype == F32 and wl.dtype == F32 and wm.numel() % 4 == 0 and bm.numel() % 4 == 0)
        if ok and getattr(self, "param_flat", None) is not None and wm.requires_grad:
            # the parameters are views of the sharded flat buffer and must stay there: fuse only where they lie
            # back to back in it (they do whenever the arena could glue them)
            ok = self._back_to_back(wm.data, wl.data) and self._back_to_back(bm.data, bl.data)
        return ok

    def _fuse_heads(self):
        """Make (or keep) the two heads' parameters views of the stacked buffers; returns (W [2L,H], b [2L])."""
        wm, wl = (self.p[h + ".weight"] for h in self.HEADS)
        bm, bl = (self.p[h + ".bias"] for h in self.HEADS)
        L, H = wm.shape
        W = self._ws.get("heads.W")
        b = self._ws.get("heads.b")
        ok = (W is not None and tuple(W.shape) == (2 * L, H) and wm.data_ptr() == W.data_ptr()
              and wl.data_ptr() == W[L:].data_ptr() and bm.data_ptr() == b.data_ptr()
              and bl.data_ptr() == b[L:].data_ptr())
        if not ok:
            if self._back_to_back(wm.data, wl.data) and self._back_to_back(bm.data, bl.data):
                # already adjacent in one storage (the sharded flat parameter buffer): views, nothing moves
                W = wm.detach().as_strided((2 * L, H), (H, 1), wm.storage_offset())
                b = bm.detach().as_strided((2 * L,), (1,), bm.storage_offset())
            else:
                W = torch.cat([wm.detach(), wl.detach()], 0).contiguous()
                b = torch.cat([bm.detach(), bl.detach()], 0).contiguous()
                wm.data, wl.data = W[:L], W[L:]
                bm.data, bl.data = b[:L], b[L:]
            self._ws["heads.W"], self._ws["heads.b"] = W, b
            self._wplanes.pop(self.HEADW, None)
        return W, b

    def _heads_version(self):
        return tuple(self.p[h + ".weight"]._version for h in self.HEADS)

    # ------------------------------------------------------------------ helpers
    # bf16 mode, narrow (fp32-operand) GEMMs: both operands are split ONCE into bf16 planes (hi, lo) and multiplied
    # as three bf16 products by the LDS-DMA kernel (C2_BF16X3P); shapes it cannot take stay on the in-kernel split.
    @staticmethod
    def _r8(n):
        """Padded length of a plane operand's rows (and of a weight's row count): the next multiple of 8, and at least the 64
        of one K-tile -- the LDS-DMA kernels take reductions of 64 and more, and a shorter one (the reference's latent
        widths 26 / 10, a 30-way classifier) runs over its zero pad like any other padded length."""
        return max(64, (int(n) + 7) // 8 * 8)

    def _planes_of(self, t):
        """(hi, lo) bf16 planes of this step's fp32 matrix t [rows, cols].  A width that is not a multiple of 8 gets planes
        with the row pitch rounded up and zero pad columns (the LDS-DMA kernels move 16-byte chunks from 16-byte aligned
        rows; a reduction over the padded width adds exact zeros): the [2, rows, cols] view of that storage is returned."""
        key = (t.data_ptr(), tuple(t.shape))
        pl = self._aplanes.get(key)
        if pl is None:
            rows, cols = t.shape
            c8 = self._r8(cols)
            full = self.buf(("pl", len(self._aplanes)), (2, rows, c8), torch.bfloat16, zero=c8 != cols)
            pl = ops.split_bf16(t, full[:, :, :cols] if c8 != cols else full)
            self._aplanes[key] = pl
        return pl

    def _tail_planes(self, key, t, pitched=True):
        """bf16x3 mode: planes buffer that the layer-tail kernel fills together with its fp32 output `t` (registered
        as t's planes for this step), or None when t is not an fp32 operand of a narrow GEMM / shape not vectorisable.
        A width that is not a multiple of 8 (or below 64) gets the [2, B, H] view of zero-padded storage, as _planes_of
        makes (pitched=False: dense planes or none -- the BatchNorm tails)."""
        B, H = t.shape
        if not (self.x3 and t.dtype == F32 and t.is_contiguous() and H % 4 == 0 and H <= 1024 and (B * H) % 4 == 0):
            return None  # (_planes_of splits into padded storage when a GEMM asks)
        c8 = self._r8(H)
        if c8 != H and not pitched:
            return None
        full = self.buf(key, (2, B, c8), torch.bfloat16, zero=c8 != H)
        pl = full[:, :, :H] if c8 != H else full
        self._aplanes[(t.data_ptr(), (B, H))] = pl
        return pl

    def _wplane_storage(self, shape, old):
        """[2, N, K] planes of a weight, kept from `old` when that has the shape; a new one is a view of zero-filled
        [2, N8, K8] storage when N or K is not a multiple of 8 (zero pad columns AND rows: the input gradient reduces
        over the weight's rows)."""
        if old is not None and old.shape[1:] == tuple(shape):
            return old
        N, K = shape
        n8, k8 = self._r8(N), self._r8(K)
        if (n8, k8) == (N, K):
            return None  # (split_bf16 makes a dense one)
        return torch.zeros((2, n8, k8), dtype=torch.bfloat16, device=self.dev)[:, :N, :K]

    def _wplanes_of(self, name):
        if name == self.HEADW:
            W, ver = self._ws["heads.W"], self._heads_version()
            c = self._wplanes.get(name)
            if c is None or c[0] != ver or c[1].shape[1:] != W.shape:
                c = (ver, ops.split_bf16(W, self._wplane_storage(W.shape, c[1] if c is not None else None)))
                self._wplanes[name] = c
            return c[1]
        p = self.p[name]
        c = self._wplanes.get(name)
        if c is None or c[0] != p._version or c[1].shape[1:] != p.shape:
            c = (p._version, ops.split_bf16(p.detach(), self._wplane_storage(p.shape, c[1] if c is not None else None)))
            self._wplanes[name] = c
        return c[1]

    def _planes_fit(self, mode, M, N, R):
        """Can the [M, N] GEMM reduced over R run on bf16 planes, and over which reduction length?  mode: "fwd" (both
        operands reduced along their columns), "dgrad" (the first along its columns, the weight along its rows), "wgrad" (both
        along their rows: the batch).  Returns the (padded) reduction length, or 0."""
        r8 = self._r8
        if not self.x3 or ops.gemm_is_tiny(M, N, R):
            return 0
        if mode == "wgrad":
            if R % 8 or R < 64:
                return 0
            R8, lds = R, (r8(M), r8(N))
        else:
            R8 = r8(R)
            lds = (R8, R8) if mode == "fwd" else (R8, r8(N))
        return R8 if ops.planes_ok(M, N, R8, lds) else 0

    def _pair(self, x, y, M, N, R, wx=None, wy=None, mode="fwd"):
        """Operand pair of one GEMM ([M,N] output reduced over R): planes when both are fp32 in bf16x3 mode and the
        shape allows, else unchanged.  wx / wy: parameter name when that operand is a weight; mode: see _planes_fit.
        Returns (x, y, planes).  Where the reduction length is not a multiple of 8 the planes are handed over widened
        over their zero pad (columns; the weight's rows in "dgrad")."""
        if not (x.dtype == F32 and y.dtype == F32 and x.is_contiguous() and y.is_contiguous()):
            return x, y, False
        R8 = self._planes_fit(mode, M, N, R)
        if not R8:
            return x, y, False
        px = self._wplanes_of(wx) if wx else self._planes_of(x)
        py = self._wplanes_of(wy) if wy else self._planes_of(y)
        if R8 != R:
            cols = lambda pl: pl.as_strided((2, pl.shape[1], R8), pl.stride(), pl.storage_offset())  # noqa: E731
            rows = lambda pl: pl.as_strided((2, R8, pl.shape[2]), pl.stride(), pl.storage_offset())  # noqa: E731
            assert px.stride(1) >= R8 and (py.stride(1) >= R8 if mode == "fwd" else py.stride(0) >= R8 * py.stride(1))
            px = cols(px)
            py = cols(py) if mode == "fwd" else rows(py)
        return px, py, True

    def _fwd(self, a, wname, y, bias=None):
        """y = a @ W^T (+ bias), unsplit."""
        a, w = self._widened(a, self.weight(wname))
        a2, w2, _ = self._pair(a, w, a.shape[0], w.shape[0], w.shape[1], wy=wname)
        ops.linear_fwd(a2, w2, y, bias=bias, x3=self.x3)

    def _widened(self, a, w):
        """Operands of y = a @ w^T that are both zero-padded along the reduction (abuf / weight() / the dataset's resident
        encoder input at a width that is not a multiple of 8): views over the padded width, so that the reduction runs in
        16-byte chunks and adds exact zeros."""
        la, lw = self._padded_width(a), self._padded_width(w)
        if la and la == lw:
            a = a.as_strided((a.shape[0], la), (la, 1), a.storage_offset())
            w = w.as_strided((w.shape[0], lw), (lw, 1), w.storage_offset())
        return a, w

    def _dgrad(self, dy, wname, out_slabs, alpha=None):
        w = self.weight(wname)
        ld = self._padded_width(dy)
        if ld and w.dtype == dy.dtype and w.shape[0] == dy.shape[1] and w.stride(1) == 1 and w.stride(0) % 8 == 0 \
                and w.untyped_storage().nbytes() >= (w.storage_offset() + ld * w.stride(0)) * 2:
            # reduction over the genes' padded count: dy's pad columns and w's pad rows are zero
            dy = dy.as_strided((dy.shape[0], ld), (ld, 1), dy.storage_offset())
            w = w.as_strided((ld, w.shape[1]), (w.stride(0), 1), w.storage_offset())
        d2, w2, _ = self._pair(dy, w, dy.shape[0], w.shape[1], w.shape[0], wy=wname, mode="dgrad")
        ops.linear_dgrad(d2, w2, out_slabs, nslab=out_slabs.shape[0], alpha=alpha, x3=self.x3)

    def _dgrad_nslab(self, dy, wname):
        w = self.weight(wname)
        M, K, R = dy.shape[0], w.shape[1], w.shape[0]
        pl = dy.dtype == F32 and w.dtype == F32 and bool(self._planes_fit("dgrad", M, K, R))
        if not pl and self._padded_width(dy):
            R = self._padded_width(dy)  # (_dgrad reduces a 2-byte pair over the padded length)
        return ops.pick_nslab(M, K, R, dy.dtype, self.x3, planes=pl)

    def _linear_slabs(self, key, a, wname, B):
        w = self.weight(wname)
        if isinstance(a, tuple):  # (resident bf16 matrix, int32 row index): the batch is read in place
            af, w = self._widened(a[0], w)
            N, K = w.shape
            ns = ops.pick_nslab(B, N, K, w.dtype, self.x3)
            y = self.buf(key + ".Y", (ns, B, N))
            ops.linear_fwd_rows(af, a[1], B, w, y, nslab=ns)
            return y
        a, w = self._widened(a, w)
        N, K = w.shape
        a2, w2, pl = self._pair(a, w, B, N, K, wy=wname)
        ns = ops.pick_nslab(B, N, K, a.dtype, self.x3, planes=pl)
        y = self.buf(key + ".Y", (ns, B, N))
        ops.linear_fwd(a2, w2, y, nslab=ns, x3=self.x3)
        return y

    def _wgrad(self, dy, a, name, alpha=None, g=None, rank1=None):
        """grad[name] = alpha * dy^T a (+ r (x) c for rank1 = (r [N], c [K]), added in the GEMM's epilogue); a
        short-and-wide weight gradient splits the batch reduction into slabs so that the launch fills the chip, then
        sums them (deterministic)."""
        if g is None:
            g = self.gbuf(name)
        N, K = g.shape
        if isinstance(a, tuple):  # (resident bf16 matrix, int32 row index): gene-width by construction
            dy = self._narrow_t(dy, "dyt", N, K, a[0])  # (the narrow operand transposed: see _narrow_t)
            if not self._wgrad_two_part(dy, a, g, alpha, rank1):
                ns = ops.pick_nslab(N, K, dy.shape[0], dy.dtype, self.x3)
                if ns == 1:
                    self._wg(dy, a, g, alpha=alpha, rank1=rank1)
                else:
                    slabs = self.buf(f"wgrad.slabs.{ns}x{N}x{K}", (ns, N, K))
                    self._wg(dy, a, slabs, nslab=ns, alpha=alpha, rank1=rank1)
                    ops.sum_slabs(slabs, g)
            return
        dy2, a2, pl = self._pair(dy, a, N, K, dy.shape[0], mode="wgrad")
        if not pl:
            at = self._narrow_t(a, "at", N, K, dy)
            if self._wgrad_two_part(dy, at, g, alpha, rank1):
                return
            if isinstance(at, _T):
                ns = ops.pick_nslab(N, K, dy.shape[0], dy.dtype, self.x3)
                if ns == 1:
                    self._wg(dy, at, g, alpha=alpha, rank1=rank1)
                else:
                    slabs = self.buf(f"wgrad.slabs.{ns}x{N}x{K}", (ns, N, K))
                    self._wg(dy, at, slabs, nslab=ns, alpha=alpha, rank1=rank1)
                    ops.sum_slabs(slabs, g)
                return
        ns = ops.pick_nslab(N, K, dy.shape[0], dy.dtype, self.x3, planes=pl)
        if ns == 1:
            ops.linear_wgrad(dy2, a2, g, alpha=alpha, x3=self.x3, rank1=rank1)
        elif self._slab_jobs is not None and g.is_contiguous():
            # a narrow layer inside a backward pass: its slabs wait (in a buffer of their own) for the one launch that
            # closes all of them (_flush_slab_sums)
            slabs = self.buf(f"wgrad.slabs.{name or 'heads'}.{ns}x{N}x{K}", (ns, N, K))
            ops.linear_wgrad(dy2, a2, slabs, nslab=ns, alpha=alpha, x3=self.x3, rank1=rank1)
            self._slab_jobs.append((slabs, g))
        else:
            slabs = self.buf(f"wgrad.slabs.{ns}x{N}x{K}", (ns, N, K))
            ops.linear_wgrad(dy2, a2, slabs, nslab=ns, alpha=alpha, x3=self.x3, rank1=rank1)
            ops.sum_slabs(slabs, g)

    def _flush_slab_sums(self):
        """Close the split reductions that wait in self._slab_jobs with one launch (c2_sum_slabs_multi)."""
        if self._slab_jobs:  # (None outside a backward pass)
            ops.sum_slabs_multi(self._slab_jobs)
            self._slab_jobs.clear()

    def _narrow_t(self, t, key, N, K, other):
        """A gene-width bf16 weight gradient [N, K] reduced over the batch multiplies faster when its NARROW operand
        ([B, H]: 8 MB at 4096 x 1024) is handed over transposed -- the cells contiguous, read row-wise, 16x16x32 MFMA -- than
        with both operands through the transposing LDS read (c2_linear_wgrad_bt / _rows_at; same result bit for bit).
        Returns the transposed copy wrapped in _T when the 256x256 kernel takes the shape, else `t` unchanged.
        `other`: the gene-width operand -- those entry points refuse what the LDS-DMA kernel cannot read (rows that do not
        start on 16 bytes: a gene count that is not a multiple of 8), where c2_linear_wgrad falls back to the
        register-staged kernel by itself; so such operands stay on it."""
        B, H = t.shape
        if not (t.dtype == torch.bfloat16 and B % 8 == 0 and B >= 1024 and H <= 2048 and max(N, K) >= self.WIDE
                and other.dtype == torch.bfloat16 and other.stride(0) % 8 == 0 and other.data_ptr() % 16 == 0
                and ops.gemm_tile(t.dtype, N, K, B, self.x3) == 256):
            return t
        return _T(ops.transpose_bf16(t, self.buf(f"wgrad.{key}.{H}x{B}", (H, B), torch.bfloat16)))

    @staticmethod
    def _wg(dy, a, out, nslab=1, alpha=None, cols=None, rank1=None):
        """Weight-gradient launch for a plain activation matrix or a (resident matrix, row index) pair; cols: column
        slice of the activation.  A _T operand is the transposed copy _narrow_t made (dy for the row-indexed form, the
        activation otherwise); slices of it are taken along its rows."""
        if isinstance(a, tuple):
            af = a[0] if cols is None else a[0][:, cols]
            if isinstance(dy, _T):
                ops.linear_wgrad_rows_at(dy.t, af, a[1], out, nslab=nslab, alpha=alpha, rank1=rank1)
            else:
                ops.linear_wgrad_rows(dy, af, a[1], out, nslab=nslab, alpha=alpha, rank1=rank1)
        elif isinstance(a, _T):
            ops.linear_wgrad_bt(dy, a.t if cols is None else a.t[cols], out, nslab=nslab, alpha=alpha, rank1=rank1)
        else:
            ops.linear_wgrad(dy, a if cols is None else a[:, cols], out, nslab=nslab, alpha=alpha, rank1=rank1)

    def _wgrad_two_part(self, dy, a, g, alpha, rank1=None):
        """A gene-width weight gradient whose 256x256 tiles number a little more than a multiple of the 256 CUs
        (19000 x 1024: 300 tiles = two rounds, the second 17 % full) runs as two launches: the tiles of the full
        rounds unsplit, the remaining strip with its batch reduction split into slabs so that it is one short round."""
        N, K = g.shape
        M = dy.shape[0]
        if ops.gemm_tile(dy.dtype, N, K, M, self.x3) != 256:
            return False
        tm, tn = (N + 255) // 256, (K + 255) // 256
        full, rem = divmod(tm * tn, 256)
        per = min(tm, tn)
        if full < 1 or rem == 0 or rem > 96 or (full * 256) % per or M < 1024:
            return False
        cut = (full * 256 // per) * 256  # rows (tm >= tn) or columns of the output covered by the full rounds
        by_rows = tm >= tn
        rest = (N if by_rows else K) - cut
        # (the strip is a problem of its own: the transposed-operand and row-indexed entry points exist for the 256x256
        # kernel only and refuse what it does not take -- a strip of 1456 x 976 at 17 840 genes x 976 was below its size)
        strict = isinstance(a, (tuple, _T)) or isinstance(dy, _T)
        if strict and ops.gemm_tile(dy.dtype, rest if by_rows else N, K if by_rows else rest, M, self.x3) != 256:
            return False
        tiles2 = ((rest + 255) // 256) * per
        ns = max(1, min(16, 256 // tiles2, M // 256))
        r1 = (lambda rs, cs: None) if rank1 is None else (lambda rs, cs: (rank1[0][rs], rank1[1][cs]))  # noqa: E731
        if by_rows:
            self._wg(_cols(dy, slice(0, cut)), a, g[:cut], alpha=alpha, rank1=r1(slice(0, cut), slice(0, K)))
            slabs = self.buf(f"wgrad.strip.{ns}x{rest}x{K}", (ns, rest, K))
            self._wg(_cols(dy, slice(cut, None)), a, slabs, nslab=ns, alpha=alpha, rank1=r1(slice(cut, N), slice(0, K)))
            ops.sum_slabs(slabs, g[cut:])
        else:
            self._wg(dy, a, g[:, :cut], alpha=alpha, cols=slice(0, cut), rank1=r1(slice(0, N), slice(0, cut)))
            slabs = self.buf(f"wgrad.strip.{ns}x{N}x{rest}", (ns, N, rest))
            self._wg(dy, a, slabs, nslab=ns, alpha=alpha, cols=slice(cut, None), rank1=r1(slice(0, N), slice(cut, K)))
            ops.sum_slabs_2d(slabs, g[:, cut:])
        return True

    @contextlib.contextmanager
    def _param_side(self, level=2):
        """Queue what runs inside on the parameter-gradient stream, after everything the current stream holds so far.
        level: C2_SIDE_LEVEL at or above which this site forks (1: the output layer's gene-width gradients only)."""
        if not self.side_on or self.comm is not None or level > self.side_level:
            yield
            return
        if self._wstream is None:
            self._wstream = torch.cuda.Stream(self.dev)
        self._wstream.wait_stream(torch.cuda.current_stream(self.dev))
        self._side_used = True
        with torch.cuda.stream(self._wstream):
            yield

    def _join_side(self):
        """The current stream waits for the parameter-gradient stream (end of a backward pass)."""
        if self._side_used:
            torch.cuda.current_stream(self.dev).wait_stream(self._wstream)
            self._side_used = False

    def _batch_center(self, x, log_in):
        """Centre of a batch handed in as tensors: the per-gene mean of its own encoder input (any vector is exact; the
        mean leaves the smallest operand).  [G rounded up to 8] fp32, pad zero."""
        B, G = x.shape
        c = self.buf("enc0.center", ((G + 7) // 8 * 8,), zero=True)
        sums = ops.col_sum(x, log_in, self.buf("enc0.center.sums", (G,), torch.float64),
                           self.buf("enc0.center.ws", (ops.col_sum_ws(B, G),), torch.float64))
        if self.exchange:
            # ranks hold equal shares of the global batch; their double sums add up to the global batch's sum exactly, so
            # N ranks round the same operand values as one process on the whole batch
            self.comm.allreduce_scalars(sums)
        return ops.col_mean_from_sums(sums, G, B * self.world_size, c)

    def _set_center(self, center):
        """Adopt the centre c of this batch's encoder input (or None): the first layer's bias becomes b + W c (fp32, from the
        master weight: the centre's share of the layer must not see the bf16 rounding of W)."""
        self._center = center
        if center is None:
            return
        blk = self.enc[0]
        w = self.p[blk.w].detach()
        ops.gemv_bias(w, center, self.p[blk.b].detach() if blk.b else None, self.buf("enc0.bias_eff", (w.shape[0],)))

    def _block_fwd(self, blk, a, cond, masks, train, save):
        B = a.shape[0] if not isinstance(a, tuple) else self._B
        y = self._linear_slabs(blk.name, a, blk.w, B)
        H = y.shape[2]
        act = self.abuf(blk.name + ".act", (B, H), blk.out_t)
        yhat = self.buf(blk.name + ".yhat", (B, H)) if save else None
        rstd = self.buf(blk.name + ".rstd", (B,)) if save else None
        p = self.cfg["dr_rate"]
        mask = masks.get(blk.drop_name) if (train and p > 0) else None
        planes = self._tail_planes(blk.name + ".act.pl", act, pitched=not blk.norm)
        bias = self.p[blk.b].detach() if blk.b else None
        if blk is self.enc[0] and self._center is not None:
            bias = self._ws["enc0.bias_eff"]
        wc = self.p[blk.wc].detach() if blk.wc else None
        if blk.norm:
            self._bn_fwd(blk, y, bias, cond if blk.use_cond else None, wc, mask, p if mask is not None else 0.0, act,
                         planes, train)
            return act
        ops.epilogue_fwd(y, bias, cond if blk.use_cond else None, wc, mask, p if mask is not None else 0.0,
                         self.cfg["use_ln"], act, yhat, rstd, planes=planes)
        return act

    def _heads_fwd(self, a, B, L):
        """mu, logvar [B, L] (views): one stacked GEMM when the two heads can share it, else two."""
        self._heads_fused = self._heads_fusable()
        if self._heads_fused:
            W, b = self._fuse_heads()
            mulv = self.buf("mulv.cat", (B, 2 * L))
            self._fwd(a, self.HEADW, mulv, bias=b)
            self._mu, self._lv = mulv[:, :L], mulv[:, L:]
        else:
            mulv = self.buf("mulv", (2, B, L))
            for k, h in enumerate(self.HEADS):
                self._fwd(a, h + ".weight", mulv[k], bias=self.p[h + ".bias"].detach())
            self._mu, self._lv = mulv[0], mulv[1]
        return self._mu, self._lv

    def _bn_fwd(self, blk, y, bias, cond, wc, mask, p, act, planes, train):
        """BatchNorm1d tail (modules.py:298-302): statistics of the (global) batch in train mode, the running ones
        in eval mode.  torch.nn.BatchNorm1d.forward: running statistics move only while training AND
        track_running_stats (the trainer clears it on layers frozen by surgery, trainer.py:307-312)."""
        _, B, H = y.shape
        mod = self.bn[blk.norm]
        if mod.momentum is None:
            raise NotImplementedError("BatchNorm1d(momentum=None) (cumulative average) is outside the path")
        if mod.running_mean is None:
            raise NotImplementedError("BatchNorm1d without running-statistic buffers is outside the path")
        h = self.buf(blk.name + ".yhat", (B, H))  # pre-norm values, normalised in place by bn_fwd
        stats = self.buf(blk.name + ".bnstats", (2, H))
        part = self.buf(blk.name + ".bnpart", (1, 2, H))
        ops.bn_stats(y, bias, cond, wc, h, part[0], self.buf(f"bn.ws.{B}x{H}", (ops.bn_ws(B, H),)))
        parts, total = None, B
        if train:
            parts = part
            if self.exchange:
                parts = self.comm.allgather_stats(part[0])
                total = B * self.world_size
        self._bn_total = getattr(self, "_bn_total", {})
        self._bn_total[blk.name] = (total, bool(train))
        ops.bn_fwd(h, parts, B, total, self.p[blk.norm + ".weight"].detach(), self.p[blk.norm + ".bias"].detach(),
                   mod.running_mean, mod.running_var, mod.num_batches_tracked, mod.momentum,
                   train and mod.track_running_stats, mask, p, act, stats, planes=planes, eps=mod.eps)

    def _block_bwd(self, blk, dact, a_in, cond, masks, need_da, out_slabs=None, extra=None, extra_scale=None):
        """dact: fp32 [B,H] or slabs.  Returns the slabs of d(a_in) (or None)."""
        B = a_in.shape[0] if not isinstance(a_in, tuple) else self._B
        H = self.p[blk.w].shape[0]
        p = self.cfg["dr_rate"]
        mask = masks.get(blk.drop_name) if p > 0 else None
        dh = self.abuf(blk.name + ".dh", (B, H), blk.gt)
        planes = self._tail_planes(blk.name + ".dh.pl", dh, pitched=not blk.norm)
        if blk.norm:
            self._bn_bwd(blk, dact, mask, p if mask is not None else 0.0, dh, planes, extra, extra_scale)
        else:
            ops.epilogue_bwd(dact, mask, p if mask is not None else 0.0, self.cfg["use_ln"],
                             self._ws[blk.name + ".yhat"], self._ws[blk.name + ".rstd"], dh, extra=extra,
                             extra_scale=extra_scale, planes=planes)
        need_w = self.p[blk.w].requires_grad
        need_b = blk.b is not None and self.p[blk.b].requires_grad
        need_c = blk.wc is not None and self.p[blk.wc].requires_grad
        # the activation gradient first (the chain every later launch of this pass waits for) ...
        centred = blk is self.enc[0] and self._center is not None and need_w
        paired = False
        if need_da:
            K = self.p[blk.w].shape[1]
            if out_slabs is None:
                out_slabs = self.buf(blk.name + ".da", (self._dgrad_nslab(dh, blk.w), B, K))
            # a tiny layer (the classifier, the decoder's first layer at the reference's batch of 128): input gradient and
            # weight gradient in ONE launch
            paired = need_w and not centred and self._tiny_pair(dh, a_in, blk.w, out_slabs)
            if not paired:
                self._dgrad(dh, blk.w, out_slabs)
        # ... then the parameter gradients, which only the optimizer waits for.
        # A centred first layer: dW = dh^T (x - c) + (sum_b dh_b) (x) c -- the column sums of dh (the bias gradient) come
        # first, the rank-one term rides in the weight-gradient GEMM's epilogue
        with self._param_side():
            db = None
            if need_b or need_c or centred:
                C = cond.shape[1] if need_c else 0
                ws = self.buf("bcg.ws", (ops.bias_cond_grad_ws(B, H, C),))
                db = self.gbuf(blk.b) if need_b else (self.buf("enc0.dhsum", (H,)) if centred else None)
                ops.bias_cond_grad(dh, cond if need_c else None, db, self.gbuf(blk.wc) if need_c else None, ws)
            if need_w and not paired:
                self._wgrad(dh, a_in, blk.w, rank1=(db, self._center) if centred else None)
        return out_slabs if need_da else None

    def _tiny_pair(self, dh, a_in, wname, out_slabs):
        """Both GEMMs of a layer's backward pass in one launch when both are tiny fp32 problems with unsplit outputs
        (c2_linear_bwd_tiny); False: not such a layer, nothing launched."""
        if isinstance(a_in, (tuple, _T)) or out_slabs.shape[0] != 1:
            return False
        w = self.weight(wname)
        B, H = dh.shape
        K = w.shape[1]
        if not (dh.dtype == F32 and a_in.dtype == F32 and w.dtype == F32 and tuple(a_in.shape) == (B, K) and B > 0
                and ops.gemm_is_tiny(B, K, H) and ops.gemm_is_tiny(H, K, B)
                and ops.pick_nslab(H, K, B, F32, self.x3, planes=False) == 1):
            return False
        ops.linear_bwd_tiny(dh, w, a_in, out_slabs[0], self.gbuf(wname))
        return True

    def _partition(self, cond_all, status):
        """Partition ids of the batch's condition rows + the pair-weight table of the MMD (losses.py:202-249)."""
        cfg = self.cfg
        Bg = cond_all.shape[0]
        part = self.buf("mmd.part", (Bg,), torch.int32)
        wt = self.buf("mmd.wt", (64 * 64,))
        pws = self.buf("mmd.pws", (int(ops.lib.c2_partition_ws(Bg)),), torch.uint8)
        ops.partition(cond_all, cfg["n_conditions"], cfg["main_dataset"], cfg["mmd_boundary"], part, wt, pws,
                      status[1:2])
        return part, wt

    def _bn_bwd(self, blk, dact, mask, p, dh, planes, extra, extra_scale):
        B, H = dh.shape
        gn, bn_ = blk.norm + ".weight", blk.norm + ".bias"
        gamma, beta = self.p[gn].detach(), self.p[bn_].detach()
        total, batch_stats = self._bn_total[blk.name]
        dy = self.buf(f"bn.dy.{B}x{H}", (B, H))
        sums = self.buf(blk.name + ".bnsums", (2, H))
        ops.bn_bwd_reduce(dact, mask, p, self._ws[blk.name + ".yhat"], gamma, beta, dy, sums,
                          self.gbuf(gn) if self.p[gn].requires_grad else None,
                          self.gbuf(bn_) if self.p[bn_].requires_grad else None,
                          self.buf(f"bn.ws.{B}x{H}", (ops.bn_ws(B, H),)), extra=extra, extra_scale=extra_scale)
        if batch_stats and self.exchange:
            self.comm.allreduce_scalars(sums)  # dgamma / dbeta above stay local sums: the arena all-reduce adds them
        ops.bn_bwd(dy, self._ws[blk.name + ".yhat"], sums, total, batch_stats, gamma, self._ws[blk.name + ".bnstats"],
                   dh, planes=planes)

    # ------------------------------------------------------------------ forward
    def forward(self, batch, eps, masks, train=True, save=True, classes_given=True, defer_sums=False):
        """defer_sums: the caller hands the returned loss slots to ops.combine_losses[_dev] next, which then also sums the
        loss kernels' partials (c2_defer_loss_sums): the slots are only valid after that call."""
        defer = bool(defer_sums) and not self.exchange  # (ranks all-reduce the four sums before they are combined)
        if not defer:
            return self._forward(batch, eps, masks, train, save, classes_given, False)
        ops.defer_loss_sums(True)
        try:
            return self._forward(batch, eps, masks, train, save, classes_given, True)
        except BaseException:
            ops.drop_pending_sums()  # the sums this forward left waiting must not reach a later combine launch
            raise
        finally:
            ops.defer_loss_sums(False)

    def _forward(self, batch, eps, masks, train, save, classes_given, deferred):
        """batch: dict of device tensors x [B,G] f32, batch [B,C] i64, sizefactor [B] f32, goodloc [B,G] bool,
        classes [B,P] i64 or None.  Returns a fp32 [4] tensor (recon, kl, mmd, class_ce); leaves everything the
        backward needs in the workspace."""
        cfg = self.cfg
        x = batch.get("x")  # None: the rows live in the resident matrices (x_full / goodloc_full / row_index)
        cond = batch.get("batch")
        if x is not None:
            B, G = x.shape
        elif batch.get("xin") is not None:
            B, G = batch["xin"].shape
        else:
            B, G = batch["row_index"].numel(), batch["xin_full"].shape[1]
        Bn = B * self.world_size
        inv_B = 1.0 / Bn
        self._B, self._cond, self._masks, self._eps = B, cond, masks, eps
        self._aplanes = {}
        losses = self._ws["losses"] = self.buf("losses4", (4,))
        if not deferred:  # (the combine launch that sums the deferred partials writes all four slots)
            losses.zero_()
        # status words are STICKY: the kernels only ever write them on an error, and check_status() alone clears them,
        # so a bad batch anywhere in an epoch is still reported at the epoch's end
        status = self._ws.get("status")
        if status is None:
            status = self._ws["status"] = torch.zeros(2, dtype=torch.int32, device=self.dev)
        # every forward that saves activations for a backward gets a generation number; backward(gen=...) refuses a
        # stale one (two forwards before one backward would silently differentiate the second)
        if save:
            self.generation = getattr(self, "generation", 0) + 1
            self._build_arena()  # (sharded form: the parameters move into the flat buffer BEFORE anything reads them)
        log_in = cfg["applylog"] and cfg["recon_loss"] != "mse"
        xt = self.enc[0].gt
        ft = self.fwd16 or xt  # operand dtype of the first layer's forward (its weight gradient reads `xt`)
        pre = batch.get("xin")  # already produced by the device-resident gather
        rows = None
        wg_in = None  # what the first layer's weight gradient reads when that is not the forward's operand
        if pre is None and batch.get("xin_full") is not None:
            # the encoder input stays in the resident bf16 matrix: the first layer's GEMMs read it through the row index
            xf, r32 = batch["xin_full"], batch["row_index32"]
            H0 = self.p[self.enc[0].w].shape[0]
            if self.fwd16 is not None:
                xb = batch.get("xinb_full")
                if xb is None or xb.dtype != xt or xb.shape != xf.shape:
                    raise ValueError("an f16 xin_full needs its bf16 twin xinb_full (the weight gradient's operand)")
                wg_in = (xb, r32)
            Gr = self._padded_width(xf) or G  # (the forward reduces over the rows' zero-padded length)
            if not (xt == torch.bfloat16 and xf.dtype == ft and B % 8 == 0
                    and ops.gemm_tile(xt, B, H0, Gr) == 256 and ops.gemm_tile(xt, H0, G, B) == 256):
                raise ValueError("xin_full / row_index32 need bf16 mode and a first layer the 256x256 GEMM takes; "
                                 "gather the encoder input instead (DeviceDataset.gather(resident_xin=False))")
            rows = (xf, r32)
            xin = None
            center = batch.get("xin_center")
        elif pre is not None and pre.dtype == ft:
            xin = pre
            center = batch.get("xin_center")
            if self.fwd16 is not None:
                wg_in = batch.get("xinb")
                if wg_in is None or wg_in.dtype != xt or wg_in.shape != xin.shape:
                    raise ValueError("an f16 xin needs its bf16 twin xinb (the weight gradient's operand)")
        elif x is None:
            raise ValueError("a batch without x needs xin in the encoder's operand dtype")
        elif log_in or xt != F32:
            xin = self.abuf("xin", (B, G), ft)
            center = self._batch_center(x, log_in) if self.center_input else None
            ops.prep_input(x, xin, log_in, center)
            if self.fwd16 is not None and save:
                wg_in = self.abuf("xinb", (B, G), xt)
                ops.prep_input(x, wg_in, log_in, center)
        else:
            xin = x
            center = None
        self._xin = xin
        self._set_center(center)
        # ---- encoder
        a = xin if rows is None else rows
        self._enc_in = []
        for blk in self.enc:
            self._enc_in.append(a)
            a = self._block_fwd(blk, a, cond, masks, train, save)
        if wg_in is not None:
            self._enc_in[0] = wg_in
        self._henc = a
        L = cfg["latent"]
        mu, lv = self._heads_fwd(a, B, L)
        z32 = self.buf("z32", (B, L))
        # one partial-sum workspace per loss term: with deferred second stages (c2_defer_loss_sums) each term's partials
        # must stay untouched until the combine launch sums them
        red_ws = self.buf("red.ws", (int(ops.lib.c2_reduce_ws()),))
        red_ws_mse = self.buf("red.ws.mse", (int(ops.lib.c2_reduce_ws()),)) if cfg["recon_loss"] == "mse" else None
        red_ws_cls = self.buf("red.ws.cls", (int(ops.lib.c2_reduce_ws()),))
        zdt = self.dec[0].gt
        zt = self.buf("zt", (B, L), zdt) if zdt != F32 else z32
        ops.reparam_kl_fwd(mu, lv, eps, inv_B, cfg["use_own_kl"], z32, zt if zdt != F32 else None,
                           losses[1:2], red_ws)
        # ---- decoder
        d = zt
        self._dec_in = []
        for blk in self.dec:
            self._dec_in.append(d)
            d = self._block_fwd(blk, d, cond, masks, train, save)
        self._dlast = d
        y1 = self._ws["dec0.act"]
        # A gene count that is not a multiple of 4 with the NB likelihood on resident rows: the loss runs over G4 = G rounded
        # up, on operands that all have rows of that width -- the dataset's padded copies (DeviceDataset.resident_rows4:
        # mask 0 in the pad), logits with -inf in the pad (a softmax weight of exactly 0), dispersion rows of zeros -- so
        # that its 16-byte row walk applies (910 -> ~220 us at 4096 x 19 001); the pad genes are masked genes.
        # Batches handed in as tensors take the same route: gathered copies that kept the dataset's row pitch as they are,
        # contiguous ones through padded copies made here (two strided copies, a fifth of what the general kernel costs more).
        G4 = (G + 3) // 4 * 4
        wide4 = False
        x_nb = gl_nb = None
        if G4 != G and cfg["recon_loss"] in ("nb", "zinb") and self.rows4_on:
            if x is None:
                wide4 = all(t.stride(0) >= G4 and t.stride(0) % 4 == 0 for t in (batch["x_full"], batch["goodloc_full"]))
            else:
                x_nb, gl_nb = x, batch["goodloc"]
                if not all(t.stride(1) == 1 and t.stride(0) >= G4 and t.stride(0) % 4 == 0 for t in (x_nb, gl_nb)):
                    x_nb = self.buf("x.rows4", (B, G4), F32, zero=True)[:, :G]
                    x_nb.copy_(x)
                    gl_nb = self.buf("goodloc.rows4", (B, G4), batch["goodloc"].dtype, zero=True)[:, :G]
                    gl_nb.copy_(batch["goodloc"])
                wide4 = True
        # bf16 mode, NB likelihood at the widths its register-resident kernel takes: the logits leave the output layer's GEMM
        # as IEEE halves (2^-12 of a logit; the same mode rounds the layer's operands and dlogits to 2^-9) -- half the
        # bytes of the step's largest store, and of the loss kernel's largest read
        want_db = save and cfg["recon_loss"] == "nb" and self.p[self.head + ".bias"].requires_grad
        half_logits = (self.half_logits_on and self.head_t == torch.bfloat16 and cfg["recon_loss"] == "nb" and not wide4 and want_db
                       and d.dtype == torch.bfloat16 and ops.nb_logits_f16_ok(B, G))
        if wide4:
            fresh = ("logits4" not in self._pool) or ((B, G4), F32) not in self._pool["logits4"]
            logits4 = self.buf("logits4", (B, G4))
            if fresh:  # (the GEMM writes G columns: the pad keeps what it is given once)
                logits4[:, G:].fill_(-float("inf"))
            logits = self._ws["logits"] = logits4[:, :G]
        elif half_logits:
            logits = self.abuf("logits16", (B, G), torch.float16)  # (rows a padded pitch apart: aligned stores and loads)
            self._ws["logits"] = logits
        else:
            logits = self.buf("logits", (B, G))
        if half_logits:
            a_, w_ = self._widened(d, self.weight(self.head + ".weight"))
            ops.linear_fwd_f16out(a_, w_, logits, bias=self.p[self.head + ".bias"].detach())
        else:
            self._fwd(d, self.head + ".weight", logits, bias=self.p[self.head + ".bias"].detach())
        # ---- reconstruction loss (+ its unit-upstream gradient)
        dlogits = self.abuf("dlogits", (B, G), self.head_t)
        if wide4 and dlogits.stride(0) < G4:  # (fp32 mode: abuf pads 2-byte rows only)
            dlogits = self._ws["dlogits"] = self.buf("dlogits4", (B, G4), self.head_t, zero=True)[:, :G]
        if cfg["recon_loss"] == "nb":
            C = cfg["n_conditions"]
            if x is None:
                xs, gl, ridx = batch["x_full"], batch["goodloc_full"], batch["row_index"]
            else:
                xs, gl, ridx = (x_nb, gl_nb, None) if wide4 else (x, batch["goodloc"], None)
            # the loss kernel also sums dlogits over the batch (the output layer's bias gradient) while it has them
            theta = self.p["theta"].detach()
            if wide4:
                widen = lambda t: t.as_strided((t.shape[0], G4), (t.stride(0), 1), t.storage_offset())  # noqa: E731
                th4 = self.buf("theta4", (G4,) + tuple(theta.shape[1:]), zero=True)
                th4[:G].copy_(theta)
                dth4 = self.buf("dtheta.unit4", th4.shape)
                dbu4 = self.buf("dbias.unit4", (G4,)) if want_db else None
                ws = self.buf("nb.ws", (ops.recon_nb_ws(B, G4, C),))
                ops.recon_nb(logits4, widen(xs), widen(gl), batch["sizefactor"], th4, cond, inv_B, losses[0:1],
                             widen(dlogits), dth4, ws, status[0:1], lgx1=batch.get("lgx1"), row_index=ridx, dbias=dbu4)
                self._ws["dtheta.unit"] = dth4[:G]
                self._ws["dbias.unit"] = dbu4[:G] if want_db else None
            else:
                ws = self.buf("nb.ws", (ops.recon_nb_ws(B, G, C),))
                dbu = self.buf("dbias.unit", (G,)) if want_db else None
                self._ws["dbias.unit"] = dbu
                ops.recon_nb(logits, xs, gl, batch["sizefactor"], theta, cond, inv_B,
                             losses[0:1], dlogits, self.buf("dtheta.unit", self.p["theta"].shape), ws, status[0:1],
                             lgx1=batch.get("lgx1"), row_index=ridx, dbias=dbu)
        elif cfg["recon_loss"] == "mse":
            if x is None:
                raise ValueError("the MSE reconstruction needs the gathered x (resident_rows is an NB-only path)")
            ops.recon_mse(logits, xin if xin.dtype == F32 else x, batch["goodloc"], inv_B, losses[0:1], dlogits, red_ws_mse)
        else:  # zinb: second gene-width head for the dropout logits (modules.py:471-477, celligner2.py:286-296)
            if wide4:  # (as for NB above; the dropout logits of a pad gene are zeros, and masked)
                pi4 = self.buf("pi4", (B, G4), zero=True)
                pi = self._ws["pi"] = pi4[:, :G]
            else:
                pi = self.buf("pi", (B, G))
            self._fwd(d, self.head2 + ".weight", pi, bias=self.p[self.head2 + ".bias"].detach())
            dpi = self.abuf("dpi", (B, G), self.head_t)
            if wide4 and dpi.stride(0) < G4:
                dpi = self._ws["dpi"] = self.buf("dpi4", (B, G4), self.head_t, zero=True)[:, :G]
            C = cfg["n_conditions"]
            if x is None:
                xs, gl, ridx = batch["x_full"], batch["goodloc_full"], batch["row_index"]
            else:
                xs, gl, ridx = (x_nb, gl_nb, None) if wide4 else (x, batch["goodloc"], None)
            theta = self.p["theta"].detach()
            if wide4:
                widen = lambda t: t.as_strided((t.shape[0], G4), (t.stride(0), 1), t.storage_offset())  # noqa: E731
                th4 = self.buf("theta4", (G4,) + tuple(theta.shape[1:]), zero=True)
                th4[:G].copy_(theta)
                dth4 = self.buf("dtheta.unit4", th4.shape)
                ws = self.buf("nb.ws", (ops.recon_nb_ws(B, G4, C),))
                ops.recon_zinb(logits4, pi4, widen(xs), widen(gl), batch["sizefactor"], th4, cond, inv_B, losses[0:1],
                               widen(dlogits), widen(dpi), dth4, ws, status[0:1], row_index=ridx, lgx1=batch.get("lgx1"))
                self._ws["dtheta.unit"] = dth4[:G]
            else:
                ws = self.buf("nb.ws", (ops.recon_nb_ws(B, G, C),))
                ops.recon_zinb(logits, pi, xs, gl, batch["sizefactor"], theta, cond, inv_B,
                               losses[0:1], dlogits, dpi, self.buf("dtheta.unit", self.p["theta"].shape), ws, status[0:1],
                               row_index=ridx, lgx1=batch.get("lgx1"))
        # ---- classifier
        classes = batch.get("classes")
        self._has_cls = classes is not None
        if self._has_cls:
            c = self.as_dtype("z.cls", z32, self.cls[0].gt)
            self._cls_in = []
            for blk in self.cls:
                self._cls_in.append(c)
                c = self._block_fwd(blk, c, None, masks, train, save)
            self._clast = c
            P = cfg["n_predictors"]
            if P > 0:
                u = self.buf("u", (B, P))
                self._fwd(c, "classifier.classifier.0.weight", u, bias=self.p["classifier.classifier.0.bias"].detach())
                du = self.buf("du", (B, P), self.gemm_dtype("classifier.classifier.0.weight"))
                ops.classifier_loss(u, classes, cfg["betaclass"], losses[3:4], du, red_ws_cls)
        # ---- MMD over the (global) batch
        self._dfeat = None
        if cfg["use_mmd"]:
            feat = z32 if cfg["mmd_on"] == "z" else y1
            D = feat.shape[1]
            if self.exchange:
                feat_all, cond_all = self.comm.allgather_rows(feat, cond)
                row0 = self.rank * B
            else:
                feat_all, cond_all, row0 = feat, cond, 0
            Bg = feat_all.shape[0]
            # (running this single-workgroup kernel on a side stream beside the encoder was measured: the cross-queue
            # dependencies cost 30 us more than the 22 us they hide)
            part, wt = self._partition(cond_all, status)
            # The four-wave MMD kernel reads 16-byte feature chunks: a latent width that is not a multiple of 4 (the reference's
            # own 26, its default 10) is widened with zero columns, which change no distance and get zero gradients -- a small
            # strided copy instead of the general pair-tile kernel (182 against 95 us at 4096 x 26).
            D4 = (D + 3) // 4 * 4
            if D4 != D and D4 <= 64 and feat_all.dtype == F32 and self.rows4_on:
                wide = self.buf("mmd.feat4", (Bg, D4), zero=True)
                wide[:, :D].copy_(feat_all)
                dwide = self.buf("mmd.dfeat4", (B, D4))
                ws = self.buf("mmd.ws", (ops.mmd_ws(B, Bg, D4),))
                ops.mmd(wide, part, wt, row0, B, cfg["beta"], losses[2:3], dwide, ws)
                self._dfeat = dwide[:, :D]
            else:
                dfeat = self.buf("mmd.dfeat", (B, D))
                ws = self.buf("mmd.ws", (ops.mmd_ws(B, Bg, D),))
                ops.mmd(feat_all, part, wt, row0, B, cfg["beta"], losses[2:3], dfeat, ws)
                self._dfeat = dfeat
        if self.exchange:
            self.comm.allreduce_scalars(losses)  # every term is already normalised by the GLOBAL batch
        return losses

    # ------------------------------------------------------------------ backward
    def backward(self, g, gen=None, seed=None):
        """g: fp32 device tensor [4] = upstream gradients of (recon, kl, mmd, class_ce).
        Fills self.grad[name] for every parameter with requires_grad and returns that dict.
        gen: the generation number of the forward this backward belongs to (checked when given).
        seed = (upstream gradient of the weighted total or None, fp32 [4] device weights): g is still to be formed as
        weights * upstream -- done by the pass's first launch together with the two vectors that g[0] scales first."""
        if gen is not None and gen != getattr(self, "generation", None):
            raise RuntimeError(
                "backward of a forward whose saved activations have been overwritten: the engine keeps ONE workspace per "
                "model, so each training forward must be followed by its backward before the next training forward "
                "(sum the losses of one forward, or accumulate gradients over separate forward/backward pairs)")
        if self.comm is not None:
            self.comm.wait_deferred()  # a bucket of the previous backward still being reduced (no optimizer step since)
        cfg = self.cfg
        B, cond, masks = self._B, self._cond, self._masks
        self._touched = set()  # parameters outside this step's graph get no gradient (torch leaves .grad None)
        self._slab_jobs = []
        self._seed = seed
        try:
            return self._backward(g, cfg, B, cond, masks)
        finally:
            self._slab_jobs = None
            self._join_side()  # (also on an error: nothing may stay queued beside the caller's stream)

    def _backward(self, g, cfg, B, cond, masks):
        gr, gk, gm, gc = g[0:1], g[1:2], g[2:3], g[3:4]
        rq = lambda n: self.p[n].requires_grad  # noqa: E731
        dlogits = self._ws["dlogits"]
        G = dlogits.shape[1]
        # ---- decoder output layer
        hw, hb = self.head + ".weight", self.head + ".bias"
        # what the loss kernel left as unit-upstream sums (the output layer's bias gradient, the dispersion gradient) times gr
        unit_jobs = []
        if rq(hb) and self._ws.get("dbias.unit") is not None:
            unit_jobs.append((self._ws["dbias.unit"], self.gbuf(hb)))
        if cfg["recon_loss"] in ("nb", "zinb") and rq("theta"):
            unit_jobs.append((self._ws["dtheta.unit"], self.gbuf("theta")))
        if self._seed is not None:
            ops.scale4_dev_scale2(self._seed[0], self._seed[1], g, unit_jobs)
            self._seed = None
        else:
            for src, dst in unit_jobs:
                ops.scale(src, gr, dst)
        H1 = self.p[hw].shape[1]
        ns1 = self._dgrad_nslab(dlogits, hw)
        ns2 = self._dgrad_nslab(dlogits, self.head2 + ".weight") if self.head2 else 0
        dd = self.buf("dhead.da", (ns1 + ns2, B, H1))
        self._dgrad(dlogits, hw, dd[:ns1], alpha=gr)
        if self.head2:  # ZINB: the dropout head's input gradient lands in further slabs of the same sum
            dpi = self._ws["dpi"]
            h2w, h2b = self.head2 + ".weight", self.head2 + ".bias"
            self._dgrad(dpi, h2w, dd[ns1:], alpha=gr)
        with self._param_side(1):
            if rq(hw):
                self._wgrad(dlogits, self._dlast, hw, alpha=gr)
            if rq(hb) and self._ws.get("dbias.unit") is None:  # (NB: formed by the loss kernel, scaled above)
                ws = self.buf("bcg.ws.G", (ops.bias_cond_grad_ws(B, G, 0),))
                ops.bias_cond_grad(dlogits, None, self.gbuf(hb), None, ws, alpha=gr)
            if self.head2:
                if rq(h2w):
                    self._wgrad(dpi, self._dlast, h2w, alpha=gr)
                if rq(h2b):
                    ws = self.buf("bcg.ws.G", (ops.bias_cond_grad_ws(B, G, 0),))
                    ops.bias_cond_grad(dpi, None, self.gbuf(h2b), None, ws, alpha=gr)
        self._build_arena()
        self._reduce_bucket(0)  # overlaps everything below
        # ---- decoder hidden blocks, then the first block (mmd_on='y' enters at its output)
        L = cfg["latent"]
        n_cls_slab = 1 if (self._has_cls and cfg["n_predictors"] > 0) else 0
        for i in range(len(self.dec) - 1, 0, -1):
            dd = self._block_bwd(self.dec[i], dd, self._dec_in[i], cond, masks, True)
        # d(z): slabs of the decoder-first-layer dgrad (a short-and-narrow GEMM: split so that it fills the chip)
        # followed by one slab from the classifier branch; the reparameterisation backward sums them
        H0 = self.p[self.dec[0].w].shape[0]
        nz = ops.pick_nslab(B, L, H0, self.dec[0].gt, self.x3,
                            planes=self.dec[0].gt == F32 and bool(self._planes_fit("dgrad", B, L, H0)))
        dz_all = self.buf("dz.all", (nz + n_cls_slab, B, L))
        on_y = cfg["use_mmd"] and cfg["mmd_on"] != "z"
        self._block_bwd(self.dec[0], dd, self._dec_in[0], cond, masks, True, out_slabs=dz_all[0:nz],
                        extra=self._dfeat if on_y else None, extra_scale=gm if on_y else None)
        # ---- classifier branch
        if n_cls_slab:
            du = self._ws["du"]
            cw, cb = "classifier.classifier.0.weight", "classifier.classifier.0.bias"
            dc = self.buf("dcls.da", (1, B, self._clast.shape[1]))
            self._dgrad(du, cw, dc, alpha=gc)
            with self._param_side():
                if rq(cw):
                    self._wgrad(du, self._clast, cw, alpha=gc)
                if rq(cb):
                    ws = self.buf("bcg.ws.P", (ops.bias_cond_grad_ws(B, du.shape[1], 0),))
                    ops.bias_cond_grad(du, None, self.gbuf(cb), None, ws, alpha=gc)
            for i in range(len(self.cls) - 1, 0, -1):
                dc = self._block_bwd(self.cls[i], dc, self._cls_in[i], None, masks, True)
            self._block_bwd(self.cls[0], dc, self._cls_in[0], None, masks, True, out_slabs=dz_all[nz:nz + 1])
        # ---- reparameterisation + KL
        on_z = cfg["use_mmd"] and cfg["mmd_on"] == "z"
        mu, lv = self._mu, self._lv
        henc = self._henc
        Hk = henc.shape[1]
        inv_Bg = 1.0 / (B * self.world_size)
        xz, xs = (self._dfeat, gm) if on_z else (None, None)
        if self._heads_fused:
            dml = self.buf("dmulv.cat", (B, 2 * L))
            ops.reparam_kl_bwd(dz_all, mu, lv, self._eps, inv_Bg, cfg["use_own_kl"], gk, dml[:, :L], dml[:, L:],
                               extra=xz, extra_scale=xs)
            wn, bn_ = [h + ".weight" for h in self.HEADS], [h + ".bias" for h in self.HEADS]
            da = self.buf("dheads.da.cat", (self._dgrad_nslab(dml, self.HEADW), B, Hk))
            self._dgrad(dml, self.HEADW, da)
            with self._param_side():
                if rq(wn[0]):  # one [2L, H] view over the two (adjacent) gradient tensors
                    g0 = self.gbuf(wn[0])
                    self.gbuf(wn[1])
                    g = self.grad_flat[g0.storage_offset():g0.storage_offset() + 2 * L * Hk].view(2 * L, Hk)
                    self._wgrad(dml, henc, None, g=g)
                if rq(bn_[0]):
                    b0 = self.gbuf(bn_[0])
                    self.gbuf(bn_[1])
                    ws = self.buf("bcg.ws.2L", (ops.bias_cond_grad_ws(B, 2 * L, 0),))
                    ops.bias_cond_grad(dml, None, self.grad_flat[b0.storage_offset():b0.storage_offset() + 2 * L], None, ws)
        else:
            dml = self.buf("dmulv", (2, B, L), self.gemm_dtype("encoder.mean_encoder.weight"))
            ops.reparam_kl_bwd(dz_all, mu, lv, self._eps, inv_Bg, cfg["use_own_kl"], gk, dml[0], dml[1], extra=xz,
                               extra_scale=xs)
            da = self.buf("dheads.da", (2, B, Hk))
            for k, nm in enumerate(self.HEADS):
                self._dgrad(dml[k], nm + ".weight", da[k:k + 1])
            with self._param_side():
                for k, nm in enumerate(self.HEADS):
                    if rq(nm + ".weight"):
                        self._wgrad(dml[k], henc, nm + ".weight")
                    if rq(nm + ".bias"):
                        ws = self.buf("bcg.ws.L", (ops.bias_cond_grad_ws(B, L, 0),))
                        ops.bias_cond_grad(dml[k], None, self.gbuf(nm + ".bias"), None, ws)
        # ---- encoder blocks
        need = self._first_trainable_enc()
        for i in range(len(self.enc) - 1, -1, -1):
            if i < need:
                break
            if i == 0:
                self._reduce_bucket(1)  # overlaps the gene-width wgrad of encoder L0
            da = self._block_bwd(self.enc[i], da, self._enc_in[i], cond, masks, need_da=(i > need))
        with self._param_side():
            self._flush_slab_sums()
        self._join_side()
        if need > 0:
            self._reduce_bucket(1)
        self._reduce_bucket(2)
        if self.comm is not None:
            self.comm.wait_all()
        return {n: self.grad[n] for n in self._touched}

    def _first_trainable_enc(self):
        for i, blk in enumerate(self.enc):
            names = [blk.w] + ([blk.b] if blk.b else []) + ([blk.wc] if blk.wc else [])
            names += [blk.norm + ".weight", blk.norm + ".bias"] if blk.norm else []
            if any(self.p[n].requires_grad for n in names):
                return i
        return len(self.enc)

    # ------------------------------------------------------------------ inference pieces
    def encode(self, x, cond, eps=None):
        """Forward-only encoder: returns (mu, logvar, z) fp32 (celligner2.py:376-409)."""
        cfg = self.cfg
        B, G = x.shape
        center = None
        if cfg["applylog"] or self.enc[0].gt != F32:
            # (an MSE model trains on x but get_latent takes log(1 + x), celligner2.py:398-399: its engine has no f16 form)
            xin = self.abuf("xin", (B, G), self.fwd16 or self.enc[0].gt)
            center = self._batch_center(x, cfg["applylog"]) if self.center_input else None
            ops.prep_input(x, xin, cfg["applylog"], center)  # get_latent applies the log for every recon loss (:398-399)
        else:
            xin = x
        self._set_center(center)
        self._aplanes = {}
        a = xin
        for blk in self.enc:
            a = self._block_fwd(blk, a, cond, {}, False, False)
        L = cfg["latent"]
        mu, lv = self._heads_fwd(a, B, L)
        z = None
        if eps is not None:
            z = self.buf("z32", (B, L))
            scratch = self.buf("losses.scratch", (1,))
            scratch.zero_()
            ops.reparam_kl_fwd(mu, lv, eps, 1.0 / B, cfg["use_own_kl"], z, None, scratch,
                               self.buf("red.ws", (int(ops.lib.c2_reduce_ws()),)))
        return mu, lv, z

    def classify(self, z):
        self._aplanes = {}
        c = self.as_dtype("z.cls", z, self.cls[0].gt)
        for blk in self.cls:
            c = self._block_fwd(blk, c, None, {}, False, False)
        P = self.cfg["n_predictors"]
        u = self.buf("u", (z.shape[0], P))
        if P > 0:
            self._fwd(c, "classifier.classifier.0.weight", u, bias=self.p["classifier.classifier.0.bias"].detach())
        return u

    def decode(self, z, cond):
        """Forward-only decoder (celligner2.py:411-420): returns (output, y1) like Decoder.forward."""
        self._aplanes = {}
        d = self.as_dtype("z.dec", z, self.dec[0].gt)
        for blk in self.dec:
            d = self._block_fwd(blk, d, cond, {}, False, False)
        B = z.shape[0]
        G = self.cfg["input_dim"]
        logits = self.buf("logits", (B, G))
        self._fwd(d, self.head + ".weight", logits, bias=self.p[self.head + ".bias"].detach())
        if self.head2:
            self._fwd(d, self.head2 + ".weight", self.buf("pi", (B, G)), bias=self.p[self.head2 + ".bias"].detach())
        return logits, self._ws["dec0.act"]

    def check_status(self):
        """Host sync: raise where the reference would have raised."""
        st = self._ws.get("status")
        if st is None:
            return
        s = st.tolist()
        if s[0] or s[1]:
            st.zero_()
        if s[0] == 2:
            raise ValueError("NB dispersion exp(theta) below 1e-17 (or not a number): outside the loss kernel's domain")
        if s[0]:
            raise IndexError("condition values outside {0,1} reached the NB dispersion one-hot "
                             "(the reference's scatter_ raises 'index out of range' here)")
        if s[1]:
            raise RuntimeError("MMD partition: unsupported condition rows (values outside {0,1}, more than 64 "
                               "distinct rows, an empty main dataset or fewer partitions than mmd_boundary needs)")
