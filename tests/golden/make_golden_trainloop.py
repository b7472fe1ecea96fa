#!/usr/bin/env python3
"""Golden vectors of FIVE STEPS OF THE REFERENCE'S OWN TRAINING LOOP: the imported VQVAE in training mode,
the loop body of python/training.py:136-164 in fp32 (no autocast / GradScaler), with the optimizer and schedule the reference
constructs — torch.optim.AdamW(lr 1e-4, weight_decay 1e-4, betas (0.9, 0.999)) (training.py:98) and
torch.optim.lr_scheduler.CosineAnnealingLR stepped once per batch (training.py:101,160) — on fixed synthetic batches of 64 leaves.

Runs only in the build container (needs /root/reference and CPU torch).  Nothing of the reference is copied: the file holds
numbers the reference produced.

    python tests/golden/make_golden_trainloop.py [out.npz]      ->  tests/golden/golden_trainloop_v1.npz

Contents:
  * seeds i64 [5]                  synth.make_leaves(64, seed) regenerates batch s; seeds[0] is the first seed >= 8100 whose fp64
                                   forward keeps every ReLU input >= RELU_MARGIN from zero and assigns the fp32 run's codes (only step 1's
                                   gradients are compared element-wise; a ReLU mask that flips between two fp32 evaluations is a
                                   discontinuity no bar on rounding covers — tests/test_gpu_fulltrain.py)
  * t_max, relu_margin
  * per step s = 0..4:  s<s>/loss, mse, l1, vq_loss, perplexity (values of that step's forward), lr (the rate that step's update used),
                        s<s>/embedding f32 [256,128], s<s>/cluster_size f32 [256], s<s>/embed_avg_norm (after the step's EMA update)
  * g32/<tensor>, g64/<tensor>     FULL gradient of step 1 for six tensors (first conv, one 16->16 conv, down, proj, stem, up_conv):
                                   the reference's fp32 autograd, and the same model converted to fp64 (stored rounded to fp32)
  * gsum/<tensor> f64 [3]          (sum, L2 norm, max |.|) of step 1's fp32 gradient, every trainable tensor
  * p5/<tensor>                    the six tensors after step 5;  psum/<tensor> f64 [3] (sum, L2 norm, max |.|) for every tensor
"""
import copy
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)
sys.path.insert(0, "/root/reference/python")

from vqvdb_amd import synth  # noqa: E402
from make_golden_regimes import model_from  # noqa: E402  (imports the reference's VQVAE)

torch.set_num_threads(8)
STEPS, BATCH, T_MAX = 5, 64, 20
RELU_MARGIN = 2e-6
SIX = ["encoder.pre.0.weight", "encoder.pre.3.conv1.weight", "encoder.down.weight", "encoder.proj.weight",
       "decoder.stem.0.weight", "decoder.up_conv.weight"]


def loss_of(m, x):
    z, recon, vq_loss, ppl = m(x)                                                       # training.py:143
    mse, l1 = F.mse_loss(recon, x), F.l1_loss(recon, x)                                 # :146-147
    return 0.8 * mse + 0.2 * l1 + vq_loss, mse, l1, vq_loss, ppl                        # :150-155


def relu_safe_seed(w):
    """First seed whose fp64 forward keeps every ReLU input RELU_MARGIN away from zero and whose fp32 forward assigns the same codes."""
    real_relu, seen = F.relu, []

    def spy(t, *a, **k):
        seen.append(float(t.detach().abs().min()))
        return real_relu(t, *a, **k)
    for seed in range(8100, 8100 + 2000):
        x = torch.from_numpy(synth.make_leaves(BATCH, seed=seed)).view(-1, 1, 8, 8, 8)
        m64 = model_from(w).double().eval()
        seen.clear()
        F.relu = spy
        try:
            with torch.no_grad():
                idx64 = m64.encode(x.double())
                m64.decode(idx64)
        finally:
            F.relu = real_relu
        if min(seen) < RELU_MARGIN:
            continue
        m32 = model_from(w).eval()
        with torch.no_grad():
            if torch.equal(m32.encode(x), idx64):
                return seed, min(seen)
    raise RuntimeError("no ReLU-safe batch found")


def main(out_path=None):
    w = synth.make_weights(0)
    seed0, margin = relu_safe_seed(w)
    seeds = [seed0] + [8000 + s for s in range(1, STEPS)]
    print(f"step-1 batch: seed {seed0} (closest ReLU input to zero in fp64: {margin:.2e})")
    out = {"seeds": np.array(seeds, dtype=np.int64), "t_max": np.int64(T_MAX), "relu_margin": np.float64(RELU_MARGIN)}
    m = model_from(w)
    m.train()
    opt = torch.optim.AdamW(m.parameters(), lr=1e-4, weight_decay=1e-4, betas=(0.9, 0.999))          # training.py:98
    sched = torch.optim.lr_scheduler.CosineAnnealingLR(opt, T_max=T_MAX)                            # training.py:101
    for s in range(STEPS):
        x = torch.from_numpy(synth.make_leaves(BATCH, seed=seeds[s])).view(-1, 1, 8, 8, 8)
        if s == 0:       # the same step in fp64 (the reference model, .double()): what both fp32 evaluations are rounding noise around
            m64 = copy.deepcopy(m).double().train()
            l64 = loss_of(m64, x.double())[0]
            l64.backward()
            g64 = {n: p.grad.detach().numpy().copy() for n, p in m64.named_parameters()}
        opt.zero_grad()                                                                  # training.py:139
        loss, mse, l1, vq_loss, ppl = loss_of(m, x)
        loss.backward()                                                                  # :160 (scaler.scale(loss).backward() without the scaler)
        lr_used = opt.param_groups[0]["lr"]
        if s == 0:
            for n, p in m.named_parameters():
                g = p.grad.detach().double()
                out["gsum/" + n] = np.array([g.sum().item(), g.norm().item(), g.abs().max().item()])
                if n in SIX:
                    out["g32/" + n] = p.grad.detach().numpy().copy()
                    out["g64/" + n] = g64[n].astype(np.float32)      # the fp64 gradient rounded once (6e-8) — bars are 5e-6
            worst = max(float(np.abs(out["g32/" + n].astype(np.float64) - g64[n]).max() / np.abs(g64[n]).max()) for n in SIX)
            print(f"step 1: reference fp32 gradients vs its fp64 gradients, six tensors, relative to each tensor's max: {worst:.2e}")
        opt.step()                                                                       # :165
        sched.step()                                                                     # :167
        out[f"s{s}/loss"], out[f"s{s}/mse"], out[f"s{s}/l1"] = np.float64(loss.item()), np.float64(mse.item()), np.float64(l1.item())
        out[f"s{s}/vq_loss"], out[f"s{s}/perplexity"], out[f"s{s}/lr"] = np.float64(vq_loss.item()), np.float64(ppl.item()), np.float64(lr_used)
        out[f"s{s}/embedding"] = m.quantizer.embedding.detach().numpy().copy()
        out[f"s{s}/cluster_size"] = m.quantizer.cluster_size.detach().numpy().copy()
        out[f"s{s}/embed_avg_norm"] = np.float64(m.quantizer.embed_avg.detach().double().norm().item())
        print(f"step {s + 1}: loss {loss.item():.6f} mse {mse.item():.6f} l1 {l1.item():.6f} vq {vq_loss.item():.6f} perplexity {ppl.item():.2f} lr {lr_used:.6e}")
    for n, p in m.named_parameters():
        v = p.detach().double()
        out["psum/" + n] = np.array([v.sum().item(), v.norm().item(), v.abs().max().item()])
        if n in SIX:
            out["p5/" + n] = p.detach().numpy().copy()
    path = out_path or os.path.join(HERE, "golden_trainloop_v1.npz")
    np.savez_compressed(path, **out)
    print(f"wrote {path}: {os.path.getsize(path) / 1e6:.2f} MB, {len(out)} arrays")


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else None)
