"""What the reference does with `n_expand > 0` (runs ONLY where /root/reference exists; nothing here ships to the GPU box).

The reference lists `n_expand` as unfinished (README.md:31-43).  This probe builds its Celligner2 with n_expand = 3 in a few
configurations through tools/ref_harness.py and runs ONE forward, to record which of them the reference itself can run:
every configuration with conditions fails in Decoder.forward (modules.py:484-496: the condition columns are concatenated
to z BEFORE the last n_expand columns are split off, so FirstL_add -- CondLayers(n_expand, out, n_conditions) -- receives
n_expand columns where it splits n_conditions off again: a shape error); without conditions the NB / ZINB likelihoods have
no `batch` to form the dispersion from, and MSE runs only with the `sum(where=)` patch of SURVEY.md Appendix B.2.
DESIGN.md section 9 quotes this output."""
import contextlib
import io
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import gen_golden as gg  # noqa: E402
import ref_harness  # noqa: E402


def attempt(ns, tag, C, P, recon, patched_mse=False):
    rng = np.random.default_rng(0)
    torch.manual_seed(0)
    orig = ns.model.mse
    try:
        with contextlib.redirect_stdout(io.StringIO()):
            m = ns.Celligner2(input_dim=40, conditions=[f"c{i}" for i in range(C)], predictors=[f"p{i}" for i in range(P)],
                              hidden_layer_sizes=[16, 8], latent_dim=5, recon_loss=recon, dr_rate=0.0, n_expand=3,
                              use_mmd=False)
        m.train()
        b = gg.make_batch(rng, 12, 40, max(C, 1), P, mse=recon == "mse")
        tb = {k: torch.tensor(v) for k, v in b.items()}
        if C == 0:
            tb["batch"] = None
        if patched_mse:
            class _MaskedSum:
                def __init__(self, t):
                    self.t = t

                def sum(self, dim=-1, where=None):
                    return (self.t * where).sum(dim=dim)
            ns.model.mse = lambda r, x: _MaskedSum(orig(r, x))
        out = m(**tb)
        print(f"{tag}: runs, losses {[round(float(o), 4) for o in out]}")
    except Exception as e:  # noqa: BLE001 -- the point is to record what is raised
        print(f"{tag}: {type(e).__name__}: {str(e).splitlines()[0][:120]}")
    finally:
        ns.model.mse = orig


def main():
    ns = ref_harness.load()
    attempt(ns, "3 conditions, NB, classifier", 3, 2, "nb")
    attempt(ns, "3 conditions, MSE", 3, 0, "mse", patched_mse=True)
    attempt(ns, "1 condition, MSE", 1, 0, "mse", patched_mse=True)
    attempt(ns, "no conditions, NB", 0, 0, "nb")
    attempt(ns, "no conditions, MSE (as shipped)", 0, 0, "mse")
    attempt(ns, "no conditions, MSE (sum(where=) patched), classifier", 0, 2, "mse", patched_mse=True)


if __name__ == "__main__":
    main()
