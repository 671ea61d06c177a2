"""Input side of the training step: the tensors the reference's AnnotatedDataset + custom_collate build
(celligner2/dataset/celligner2/anndata.py:8-166, dataset/celligner2/_utils.py:39-77,
trainers/celligner2/_utils.py:79-185), kept RESIDENT on the GPU and gathered by row index with one kernel
(c2_gather_rows) instead of per-sample __getitem__ + collate + host->device copies.

Label encoding, NaN handling, size factors, the train/validation split and the stratifier weights are host
logic (numpy/pandas on the AnnData-like object) restated from the cited lines; nothing here is on the GPU
hot path except `gather`.
"""
from __future__ import annotations

import numpy as np
import torch

from . import ops


def label_encoder_2D(adata, encoder, label_sets):
    """Multi-hot label matrix [n, len(encoder)] of the obs columns `label_sets` (column name -> the values it holds):
    1 in the column of every label a row carries; a row whose label in column k is NOT in the encoder gets -1 in the
    columns of all of k's known labels (the reference's marker for "missing": dataset/celligner2/_utils.py:39-77)."""
    n = adata.shape[0]
    out = np.zeros((n, len(encoder)))
    columns = {key: np.asarray(adata.obs[key]).astype(object) for key in label_sets}
    unknown = set()
    for key, col in columns.items():
        extra = set(col) - set(encoder)
        if extra:
            unknown |= extra
            print(f"Note: obs[{key!r}] holds labels the encoder does not know ({sorted(map(str, unknown))}); "
                  "rows carrying them are marked -1 in that column's labels")
    rows = np.arange(n)
    for key, values in label_sets.items():
        known = [v for v in values if v not in unknown]
        if not known:
            continue
        where = np.array([encoder[v] for v in known], dtype=np.int64)
        slot = {v: i for i, v in enumerate(known)}
        pos = np.fromiter((slot.get(v, -1) for v in columns[key]), dtype=np.int64, count=n)  # -1: unknown label
        hit = pos >= 0
        out[rows[hit], where[pos[hit]]] = 1
        if (~hit).any():
            out[np.ix_(rows[~hit], where)] = -1
    return out


def train_test_split(adata, train_frac=0.85, cell_type_key=None):
    """Row indices (train, validation).  The cut is made per group -- the whole table, or one group per value of
    obs[cell_type_key] in order of first appearance -- after ONE `np.random.shuffle` of the group's ascending row
    indices: numpy's GLOBAL generator, consumed exactly as the reference consumes it
    (trainers/celligner2/_utils.py:79-116), so a seeded run splits the same way.  train_frac == 1: no validation set."""
    n = adata.shape[0]
    if train_frac == 1:
        return np.arange(n), None
    if cell_type_key is None:
        groups = [np.arange(n)]
    else:
        labels = adata.obs[cell_type_key]
        groups = [np.flatnonzero(np.asarray(labels == value)) for value in labels.unique().tolist()]
    kept, held = [], []
    for members in groups:
        np.random.shuffle(members)
        cut = int(np.ceil(train_frac * members.size))
        kept.append(members[:cut])
        held.append(members[cut:])
    return np.concatenate(kept), np.concatenate(held)


class DeviceDataset:
    """The reference's AnnotatedDataset with its tensors resident in HBM."""

    def __init__(self, adata, goodloc=None, condition_keys=None, condition_encoder=None, cell_type_key=None,
                 cell_type_encoder=None, predictor_encoder=None, predictor_keys=None, minweight=0.0, device="cuda"):
        self.minweight = minweight
        self.condition_keys = condition_keys
        self.condition_encoder = condition_encoder
        self.cell_type_key = cell_type_key
        self.cell_type_encoder = cell_type_encoder
        self.predictor_encoder = predictor_encoder
        self.predictor_keys = predictor_keys
        self.device = torch.device(device)
        X = np.ascontiguousarray(np.asarray(adata.X), dtype=np.float32)
        self.n, self.G = X.shape
        sf = np.asarray(adata.obs["celligner2_size_factors"], dtype=np.float32).reshape(-1)
        # counts and mask live in HBM ONCE, rows G rounded up to 4 elements apart (pad columns: count 0, mask 0), so that
        # every row starts on 16 / 4 bytes whatever the gene count and the likelihood kernels walk it in 16-byte chunks
        self.data = self._pitched(torch.from_numpy(X), torch.float32)
        self.size_factors = torch.from_numpy(sf).to(self.device)
        if goodloc is None:
            self.goodloc = self._pitched(None, torch.bool)
        else:
            self.goodloc = self._pitched(torch.as_tensor(np.ascontiguousarray(goodloc), dtype=torch.bool), torch.bool)
        self.conditions = None
        self.predictors = None
        if self.condition_keys is not None:
            sets = {key: set(adata.obs[key]) for key in self.condition_keys}
            self.conditions_np = label_encoder_2D(adata, self.condition_encoder, sets).astype(np.int64)
            self.conditions = torch.from_numpy(self.conditions_np).to(self.device)
        if self.predictor_keys is not None:
            sets = {key: set(adata.obs[key]) for key in self.predictor_keys}
            self.predictors_np = label_encoder_2D(adata, self.predictor_encoder, sets).astype(np.int64)
            self.predictors = torch.from_numpy(self.predictors_np).to(self.device)

    def __len__(self):
        return self.n

    def _pitched(self, src, dtype, chunk=16384):
        """[n, G] view over a zero-initialised [n, ops.rows_pitch(G)] device matrix holding `src` (host or device tensor;
        None: all ones), filled in row chunks (no second full-size temporary)."""
        ld = ops.rows_pitch(self.G)
        if src is not None and ld == self.G:
            return src.to(self.device, dtype).contiguous()
        full = torch.zeros((self.n, ld), dtype=dtype, device=self.device)
        view = full[:, :self.G]
        for s0 in range(0, self.n, chunk):
            if src is None:
                view[s0:s0 + chunk] = True
            else:
                view[s0:s0 + chunk] = src[s0:s0 + chunk].to(self.device, dtype)
        return view

    def _rows4_ok(self, t):
        """Rows the likelihood kernels can walk in 16-byte chunks over G rounded up to 4: a pitch that covers it and keeps
        every row on 16 (counts) / 4 (mask) bytes; the pad columns are the producer's to keep zero."""
        return t.stride(1) == 1 and t.stride(0) % 4 == 0 and t.stride(0) >= (self.G + 3) // 4 * 4

    @classmethod
    def from_tensors(cls, data, size_factors, goodloc, conditions=None, predictors=None, minweight=0.0, repitch=False):
        """Wrap tensors that already live on the GPU (synthetic benchmarks, pre-staged matrices).  [n, G] views whose rows
        are G rounded up to 4 elements apart with zero pad columns (count 0, mask 0) are used as they are.  repitch=True:
        at a gene count that is not a multiple of 4, contiguous `data` / `goodloc` are copied ONCE into such matrices and
        the dataset keeps only the copies (the caller may free its own); otherwise the fast likelihood path makes its
        padded copies on first use and the dataset holds both (resident_rows4)."""
        self = cls.__new__(cls)
        self.minweight = minweight
        self.device = data.device
        self.n, self.G = data.shape
        if repitch:
            ld = ops.rows_pitch(self.G)
            if data.stride(0) != ld:
                data = self._pitched(data, data.dtype)
            if goodloc.stride(0) != ld:
                goodloc = self._pitched(goodloc, goodloc.dtype)
        self.data, self.size_factors, self.goodloc = data, size_factors, goodloc
        self.conditions, self.predictors = conditions, predictors
        self.condition_keys = ["condition"] if conditions is not None else None
        self.predictor_keys = ["predictor"] if predictors is not None else None
        if conditions is not None:
            self.conditions_np = conditions.cpu().numpy()
        if predictors is not None:
            self.predictors_np = predictors.cpu().numpy()
        return self

    @property
    def stratifier_weights(self):
        """Sampling weight per row (dataset/celligner2/anndata.py:137-166): every label column of the condition and
        predictor matrices shares 1/(number of columns) of its matrix equally among the rows whose entry is non-zero
        (a -1 "missing" entry counts as present, as in the reference), plus `minweight`.  The per-row total is
        accumulated column by column, conditions first (a running sum, so the float64 result has the reference's bits:
        the index stream drawn from these weights is compared with the reference's)."""
        tables = []
        for name in ("conditions_np", "predictors_np"):
            m = getattr(self, name, None)
            if m is None or m.shape[1] == 0:
                continue
            present = m != 0
            count = present.sum(0)
            share = np.where(count > 0, 1.0 / (np.maximum(count, 1) * m.shape[1]), 0.0)
            tables.append(np.where(present, share[None, :], 0.0))
        if not tables:
            return np.zeros(self.n) + self.minweight
        return np.cumsum(np.concatenate(tables, 1), axis=1)[:, -1] + self.minweight

    @property
    def nb_row_const(self):
        """[n] fp32: sum_g goodloc*lgamma(x+1) per row, the data-only term of the NB likelihood (losses.py:60-64),
        evaluated once for the resident matrix (like the size factors, trainers/celligner2/_utils.py:149-152)."""
        c = getattr(self, "_nb_row_const", None)
        if c is None:
            c = self._nb_row_const = ops.nb_row_const(self.data, self.goodloc)
        return c

    def resident_rows4(self):
        """(counts, mask) as [n, G] views whose rows are G rounded up to 4 elements apart, pad columns zero: the dataset's own
        matrices (built that way by the constructor and by from_tensors(repitch=True); any matrix when G is a multiple of
        4) -- the NB loss kernel's fast path walks the rows it is pointed at in 16-byte chunks and takes a masked pad gene
        for what it is: absent.  Only a dataset wrapped around contiguous tensors at a gene count that is not a multiple of
        4 (from_tensors without repitch) makes padded copies here, once, and then holds counts and mask twice."""
        g4 = (self.G + 3) // 4 * 4
        if self._rows4_ok(self.data) and self._rows4_ok(self.goodloc):  # (a multiple of 4, or rows padded at construction)
            return self.data, self.goodloc
        c = self.__dict__.get("_rows4")
        if c is None:
            g4 = (self.G + 3) // 4 * 4
            d = torch.zeros((self.n, g4), dtype=self.data.dtype, device=self.device)
            m = torch.zeros((self.n, g4), dtype=self.goodloc.dtype, device=self.device)
            for s in range(0, self.n, 16384):
                d[s:s + 16384, :self.G] = self.data[s:s + 16384]
                m[s:s + 16384, :self.G] = self.goodloc[s:s + 16384]
            c = self.__dict__["_rows4"] = (d[:, :self.G], m[:, :self.G])
        return c

    def xin_center(self, apply_log):
        """[G rounded up to 8] fp32 (pad zero): the per-gene mean of the encoder input log(1+x) (or x) over the resident
        rows -- the vector c that bf16 mode subtracts from the encoder input before rounding it (include/c2mi.h,
        c2_col_mean; the engine adds W c to the first layer's bias and (sum dh) (x) c to its weight gradient, so the layer
        is unchanged).  Evaluated once per dataset, like the size factors."""
        cache = self.__dict__.setdefault("_xin_center", {})
        c = cache.get(bool(apply_log))
        if c is None:
            c = torch.zeros((self.G + 7) // 8 * 8, dtype=torch.float32, device=self.device)
            ops.col_mean(self.data, bool(apply_log), out=c)
            cache[bool(apply_log)] = c
        return c

    def xin_resident(self, apply_log, centered=True, dtype=torch.bfloat16):
        """[n, G] bf16 (or IEEE half): the encoder input of EVERY resident row (log(1+x) or x, minus xin_center when `centered`),
        converted once per dataset like the size factors (3.8 GB for 100k x 19k of 288 GB).  A batch is then rows of
        this matrix, read in place by the row-indexed GEMMs (c2_linear_fwd_rows / c2_linear_wgrad_rows): no per-step
        copy of the encoder input."""
        cache = self.__dict__.setdefault("_xin_res", {})
        key = (bool(apply_log), bool(centered), dtype)
        t = cache.get(key)
        if t is None:
            # (rows start on 16 bytes for the LDS-DMA GEMMs: a gene count that is not a multiple of 8 gets zero pad columns)
            t = torch.zeros((self.n, ops.pitch2(self.G)), dtype=dtype, device=self.device)[:, :self.G]
            c = self.xin_center(apply_log) if centered else None
            for s in range(0, self.n, 16384):
                ops.prep_input(self.data[s:s + 16384], t[s:s + 16384], bool(apply_log), c)
            cache[key] = t
        return t

    def gather(self, idx, xin_dtype=None, apply_log=False, bufs=None, nb_const=False, resident_rows=False,
               resident_xin=False, centered=False, xin2=False):
        """Batch dict for rows `idx` (int64 device tensor): the keys custom_collate would produce
        (trainers/celligner2/_utils.py:188-225) plus `xin`, the encoder input written by the same kernel.

        resident_rows=True (needs xin_dtype): x and goodloc are NOT copied; the batch carries the resident matrices
        and the row index instead (`x_full`, `goodloc_full`, `row_index`) and the NB loss kernel reads the rows
        where they live -- the gather then moves 6 instead of 12 bytes per gene.
        resident_xin=True (with resident_rows, bf16 input, G * n < 2^31): the encoder input is not written either:
        the batch carries `xin_full` (xin_resident) and `row_index32`, and only the per-row scalars are gathered.
        centered=True (bf16 input): xin / xin_full hold the encoder input minus `xin_center`, which the batch carries.
        xin2=True (xin_dtype float16): the batch also carries the bf16 twin of the encoder input, `xinb` / `xinb_full` (the
        first layer's forward reads IEEE halves, its weight gradient bf16)."""
        n = idx.numel()
        two_byte = xin_dtype in (torch.bfloat16, torch.float16)
        xin2 = bool(xin2) and xin_dtype == torch.float16
        centered = bool(centered) and two_byte
        center = self.xin_center(apply_log) if centered else None
        resident_rows = resident_rows and xin_dtype is not None
        resident_xin = (resident_xin and resident_rows and xin_dtype in (torch.bfloat16, torch.float16)
                        and self.n * ops.pitch2(self.G) < 2 ** 31 and n % 8 == 0)
        bufs = bufs if bufs is not None else {}

        def get(key, shape, dtype):
            t = bufs.get(key)
            if t is None or tuple(t.shape) != tuple(shape) or t.dtype != dtype:
                t = torch.empty(shape, dtype=dtype, device=self.device)
                bufs[key] = t
            return t

        def rows4(key, dtype):
            g4 = ops.rows_pitch(self.G)
            if g4 == self.G:
                return get(key, (n, self.G), dtype)
            full = bufs.get(key + ".rows4")
            if full is None or tuple(full.shape) != (n, g4) or full.dtype != dtype:
                full = bufs[key + ".rows4"] = torch.zeros((n, g4), dtype=dtype, device=self.device)
            return full[:, :self.G]

        out = dict(sizefactor=get("sizefactor", (n,), torch.float32))
        if resident_rows:
            xf, gf = self.resident_rows4()
            out.update(x_full=xf, goodloc_full=gf, row_index=idx.contiguous())
        else:
            # (gathered copies keep the padded row pitch: the batch takes the same kernels as the resident rows)
            out.update(x=rows4("x", torch.float32), goodloc=rows4("goodloc", torch.bool))
        cond_o = cls_o = None
        if self.condition_keys:
            cond_o = out["batch"] = get("batch", (n, self.conditions.shape[1]), torch.int64)
        if self.predictor_keys:
            cls_o = out["classes"] = get("classes", (n, self.predictors.shape[1]), torch.int64)
        xin = xinb = idx32 = None
        if resident_xin:
            out["xin_full"] = self.xin_resident(apply_log, centered, xin_dtype)
            if xin2:
                out["xinb_full"] = self.xin_resident(apply_log, centered, torch.bfloat16)
            idx32 = out["row_index32"] = get("row_index32", ((n + 63) // 64 * 64,), torch.int32)
        elif xin_dtype is not None:
            def two_byte_rows(key, dtype):
                g8 = ops.pitch2(self.G)
                if g8 != self.G:  # (see xin_resident)
                    full = bufs.get(key + ".padded")
                    if full is None or tuple(full.shape) != (n, g8) or full.dtype != dtype:
                        full = bufs[key + ".padded"] = torch.zeros((n, g8), dtype=dtype, device=self.device)
                    return full[:, :self.G]
                return get(key, (n, self.G), dtype)

            xin = out["xin"] = two_byte_rows("xin", xin_dtype) if two_byte else get("xin", (n, self.G), xin_dtype)
            if xin2:
                xinb = out["xinb"] = two_byte_rows("xinb", torch.bfloat16)
        rc = rc_o = None
        if nb_const:
            rc, rc_o = self.nb_row_const, get("lgx1", (n,), torch.float32)
            out["lgx1"] = rc_o
        if centered:
            out["xin_center"] = center
        ops.gather_rows(self.data, self.goodloc, self.size_factors, self.conditions if cond_o is not None else None,
                        self.predictors if cls_o is not None else None, idx, out.get("x"), out.get("goodloc"),
                        out["sizefactor"], cond_o, cls_o, xin, apply_log, rowconst=rc, rowconst_out=rc_o,
                        idx32_out=idx32, center=center if xin is not None else None, xin2_out=xinb)
        return out


def make_dataset(adata, train_frac=0.9, condition_keys=None, cell_type_key=None, condition_encoder=None,
                 cell_type_encoder=None, predictor_keys=None, predictor_encoder=None, min_weight=0.0, device="cuda"):
    """(train, validation) DeviceDatasets of an AnnData-like object, prepared as the reference prepares them
    (trainers/celligner2/_utils.py:119-185): NaN entries are remembered in the `goodloc` mask and replaced IN PLACE
    (adata.X) by the NaN-mean of their gene over all rows; the per-row sum of the imputed matrix becomes
    obs["celligner2_size_factors"]; then the split."""
    X = adata.X
    missing = np.isnan(X)
    if missing.any():
        X[missing] = np.take(np.nanmean(X, axis=0), np.nonzero(missing)[1])
    adata.obs["celligner2_size_factors"] = X.sum(1).reshape(X.shape[0], -1)
    common = dict(condition_keys=condition_keys, cell_type_key=cell_type_key, condition_encoder=condition_encoder,
                  cell_type_encoder=cell_type_encoder, predictor_encoder=predictor_encoder, predictor_keys=predictor_keys,
                  minweight=min_weight, device=device)
    parts = train_test_split(adata, train_frac, cell_type_key)
    return tuple(None if rows is None else DeviceDataset(adata[rows, :], ~missing[rows, :], **common) for rows in parts)
