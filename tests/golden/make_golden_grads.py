#!/usr/bin/env python3
"""Golden loss / gradient summaries of ONE training step of the IMPORTED reference model (python/VQVAE_v2.py VQVAE.forward in
training mode + the loss of python/training.py:147-155, fp32, no autocast) on synth.make_leaves(16, seed=5000).

Runs only in the build container.  Stores, per trainable tensor, the gradient's sum, L2 norm and first 6 values (enough to pin
a restatement; the full gradients are ~4 MB), plus the loss pieces.  No reference source is copied."""
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, HERE)
sys.path.insert(0, "/root/reference/python")
from vqvdb_amd import synth  # noqa: E402
from make_golden import build_model  # noqa: E402

torch.set_num_threads(8)


def main():
    m = build_model()
    m.train()
    x = torch.from_numpy(synth.make_leaves(16, seed=5000)).view(-1, 1, 8, 8, 8)
    z, recon, vq_loss, ppl = m(x)
    mse, l1 = F.mse_loss(recon, x), F.l1_loss(recon, x)
    loss = 0.8 * mse + 0.2 * l1 + vq_loss
    loss.backward()
    out = {"loss": np.float64(loss.item()), "mse": np.float64(mse.item()), "l1": np.float64(l1.item()), "vq_loss": np.float64(vq_loss.item())}
    for name, p in m.named_parameters():
        g = p.grad.detach().double().flatten()
        out["g:" + name] = np.concatenate([[g.sum().item(), g.norm().item()], g[:6].numpy()])
    path = os.path.join(HERE, "golden_grads_v1.npz")
    np.savez_compressed(path, **out)
    print(f"wrote {path} ({os.path.getsize(path)} B): loss {loss.item():.6f}, {sum(1 for _ in m.named_parameters())} parameter tensors")


if __name__ == "__main__":
    main()
