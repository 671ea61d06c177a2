"""End-to-end sanity on the GPU: a few epochs of the C4-shaped model (bf16 mode, every fast path on: resident rows and
encoder input, stacked heads, fused narrow GEMMs, split-operand MMD) on synthetic counts; prints the epoch logs -- the
reconstruction loss must fall and stay finite.  usage: python tools/train_sanity.py [epochs]"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import bench
from celligner2_amd.model import Celligner2
from celligner2_amd.trainer import Celligner2Trainer

def main():
    epochs = int(sys.argv[1]) if len(sys.argv) > 1 else 4
    cfg = dict(bench.CONFIGS["c4"]); cfg["N"] = 24576
    dev = torch.device("cuda", 0)
    torch.manual_seed(1)
    model = Celligner2(cfg["G"], cfg["hidden"], cfg["latent"], 0.05, False, "nb", conditions=[f"c{i}" for i in range(cfg["C"])],
                       use_mmd=True, mmd_on="z", beta=cfg["beta"], precision="bf16")
    data = bench.synth_dataset(cfg, dev)
    tr = Celligner2Trainer(model, None, train_data=data, valid_data=None, batch_size=4096, use_leiden=False, alpha_kl=1e-3,
                           monitor=False, use_early_stopping=False, rng="device", seed=1)
    tr.train(n_epochs=epochs, lr=1e-3, eps=0.01)
    for k in ("epoch_recon_loss", "epoch_kl_loss", "epoch_mmd_loss", "epoch_loss"):
        print(k, [round(float(v), 4) for v in tr.logs[k]])
    r = tr.logs["epoch_recon_loss"]
    assert all(torch.isfinite(torch.tensor(r))) and r[-1] < r[0], r
    print("ok: reconstruction loss fell from", r[0], "to", r[-1])

if __name__ == "__main__":
    main()
