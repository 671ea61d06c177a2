"""GPU diagnostic (test infrastructure): which bf16 rounding of the C4 step carries the error of the gradients that miss
1e-2 against the fp64 oracle (tests/test_benchwidth_oracle_gpu.py: log-variance head, encoder condition columns).
Runs the C4 case of that test with single roundings switched off / on:
  base        bf16 mode as shipped
  head_x3     all three GEMMs of the decoder output layer on fp32 operands (bf16x3), dlogits fp32
  enc0_x3     both GEMMs of encoder L0 on fp32 operands
  both_x3     the two above
  +dl_rounded head_x3, but the input gradient reads dlogits rounded to bf16
  +w_rounded  head_x3, but the input gradient reads the weight rounded to bf16
  +fwd_rounded head_x3, but the forward reads its operands rounded to bf16
usage: python tools/diag_bf16_c4.py [case]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from tests.test_benchwidth_oracle_gpu import CASES, _case, _reference  # noqa: E402

F32 = torch.float32


def rb(t):
    return t.to(torch.bfloat16).float()


def run(name, variant):
    from celligner2_amd import ops
    from celligner2_amd.model import Celligner2
    cfg, params, batch, eps, masks = _case(name)
    ref = _reference(name)
    c = CASES[name]
    m = Celligner2(c["G"], c["hidden"], c["L"], 0.05, False, "nb", conditions=[f"c{i}" for i in range(c["C"])],
                   use_mmd=True, mmd_on="z", beta=c["beta"], predictors=[f"p{i}" for i in range(c["P"])],
                   classifier_hidden_layer_sizes=c["cls"], precision="bf16")
    m.load_state_dict({k: torch.tensor(v) for k, v in params.items()})
    m.cuda().train()
    m._forced_noise = dict(eps=torch.tensor(eps).cuda(),
                           masks={k: torch.tensor(v.astype(np.uint8)).cuda() for k, v in masks.items()})
    eng = m._engine()
    hw = eng.head + ".weight"
    if variant.startswith("narrow16"):
        # what single-product IEEE-half narrow GEMMs would do to the gradients: the bf16x3 path on operands rounded to
        # halves first (fwd: forward GEMMs only; all: forward, input-gradient and weight-gradient GEMMs)
        h16 = lambda t: t.to(torch.float16).float()  # noqa: E731
        o_fwd, o_ls, o_dg, o_wg = eng._fwd, eng._linear_slabs, eng._dgrad, eng._wgrad
        f32 = torch.float32

        def w16(name, fn):
            if name is None or eng.gemm_dtype(name) != f32 or name == eng.HEADW:
                return fn()
            p_ = eng.p[name]
            keep = p_.data.clone()
            p_.data.copy_(h16(keep))
            eng._wplanes.pop(name, None)
            try:
                return fn()
            finally:
                p_.data.copy_(keep)
                eng._wplanes.pop(name, None)

        def ls(key, a, wname, B, _o=o_ls):
            if isinstance(a, tuple) or a.dtype != f32:
                return _o(key, a, wname, B)
            eng._aplanes.pop((a.data_ptr(), tuple(a.shape)), None)
            a2 = h16(a)
            return w16(wname, lambda: _o(key, a2, wname, B))
        eng._linear_slabs = ls
        if variant == "narrow16_all":
            def dg(dy, wname, out_slabs, alpha=None, _o=o_dg):
                if dy.dtype != f32:
                    return _o(dy, wname, out_slabs, alpha)
                s_ = 4096.0
                d2 = h16(dy * s_) / s_
                return w16(wname, lambda: _o(d2, wname, out_slabs, alpha))
            eng._dgrad = dg

            def wg(dy, a, name, alpha=None, g=None, _o=o_wg):
                if isinstance(a, tuple) or dy.dtype != f32 or a.dtype != f32:
                    return _o(dy, a, name, alpha, g)
                s_ = 4096.0
                return _o(h16(dy * s_) / s_, h16(a), name, alpha, g)
            eng._wgrad = wg
    elif variant != "base" and variant != "enc0_x3":
        eng.head_t = F32
        eng.dec[-1].out_t = F32
        gd = eng.gemm_dtype
        eng.gemm_dtype = lambda n: F32 if n == hw else gd(n)
    if variant in ("enc0_x3", "both_x3"):
        eng.enc[0].gt = F32
        gd2 = eng.gemm_dtype
        eng.gemm_dtype = lambda n: F32 if n == eng.enc[0].w else gd2(n)
    if variant in ("+dl_rounded", "+w_rounded"):
        def dgrad(dy, wname, out_slabs, alpha=None, _o=eng._dgrad):
            if wname != hw:
                return _o(dy, wname, out_slabs, alpha)
            w = eng.p[hw].detach()
            if variant == "+dl_rounded":
                dy = rb(dy)
            else:
                w = rb(w)
            ops.linear_dgrad(dy.contiguous(), w.contiguous(), out_slabs, nslab=out_slabs.shape[0], alpha=alpha, x3=True)
        eng._dgrad = dgrad
        eng._dgrad_nslab = lambda dy, wname, _o=eng._dgrad_nslab: 1 if wname == hw else _o(dy, wname)
    if variant == "+fwd_rounded":
        def fwd(a, wname, y, bias=None, _o=eng._fwd):
            if wname != hw:
                return _o(a, wname, y, bias)
            ops.linear_fwd(rb(a).contiguous(), rb(eng.p[hw].detach()).contiguous(), y, bias=bias, x3=True)
        eng._fwd = fwd
    r, k, mm, ce = m(**{k_: torch.tensor(v).cuda() for k_, v in batch.items()})
    (r + 1e-3 * k + mm + ce).backward()
    errs = {}
    for n, p in m.named_parameters():
        gr = ref["grads"].get(n)
        if gr is not None:
            errs[n] = float(np.linalg.norm(p.grad.double().cpu().numpy() - gr) / max(np.linalg.norm(gr), 1e-30))
    top = sorted(((e, n) for n, e in errs.items()), reverse=True)[:7]
    print(f"{name} {variant:13s}", "  ".join(f"{n.replace('encoder.', 'e.').replace('decoder.', 'd.')}={e:.4f}" for e, n in top), flush=True)


if __name__ == "__main__":
    case = sys.argv[1] if len(sys.argv) > 1 else "c4"
    variants = sys.argv[2:] or ["base", "head_x3", "enc0_x3", "both_x3", "+dl_rounded", "+w_rounded", "+fwd_rounded"]
    for v in variants:
        run(case, v)
