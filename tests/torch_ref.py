"""Plain PyTorch fp32 restatement of the VQ-VAE training step — TEST INFRASTRUCTURE (the gradient oracle of the HIP
backward kernels, SURVEY.md §8 f-2 stage 2).  Written from the network specification (reference
python/VQVAE_v2.py:190-275,107-156 and the loss of python/training.py:147-155), functional style over a dict of tensors
named like the reference's state_dict; autograd supplies the gradients.  Pinned to the imported reference by
tests/golden/make_golden_grads.py -> tests/test_torch_ref.py."""
from __future__ import annotations

import torch
import torch.nn.functional as F


def _rec(tape, name, t):
    """Optionally record an intermediate (and keep its gradient) under `name`."""
    if tape is not None:
        if t.requires_grad:
            t.retain_grad()
        tape[name] = t
    return t


def _gn_relu(x, w, prefix, groups):
    return F.relu(F.group_norm(x, groups, w[prefix + ".weight"], w[prefix + ".bias"], eps=1e-5))


def _res_block(x, w, prefix, groups=8, scale=0.1, tape=None, tag=""):
    t = _gn_relu(x, w, prefix + ".gn1", groups)
    y = _rec(tape, tag + ".y", F.conv3d(t, w[prefix + ".conv1.weight"], w[prefix + ".conv1.bias"], padding=1))
    u = _gn_relu(y, w, prefix + ".gn2", groups)
    return x + scale * F.conv3d(u, w[prefix + ".conv2.weight"], w[prefix + ".conv2.bias"], padding=1)


def _attention(x, w, prefix):
    m = x.mean(dim=(2, 3, 4))
    g = torch.sigmoid(F.linear(F.relu(F.linear(m, w[prefix + ".fc.0.weight"])), w[prefix + ".fc.2.weight"]))
    return x * g[:, :, None, None, None]


def encoder(x, w, tape=None):
    y1 = _rec(tape, "e.y1", F.conv3d(x, w["encoder.pre.0.weight"], w["encoder.pre.0.bias"], padding=1))
    a = _rec(tape, "e.a1", _gn_relu(y1, w, "encoder.pre.1", 4))
    a = _rec(tape, "e.a6", _res_block(a, w, "encoder.pre.3", tape=tape, tag="e.r16"))
    a = _rec(tape, "e.x7", F.conv3d(a, w["encoder.down.weight"], w["encoder.down.bias"], stride=2, padding=1))
    a = _rec(tape, "e.x11", _res_block(a, w, "encoder.res_stack.0", tape=tape, tag="e.r32"))
    a = _rec(tape, "e.x12", _attention(a, w, "encoder.attn"))
    return F.conv3d(a, w["encoder.proj.weight"], w["encoder.proj.bias"])


def pixel_shuffle3d(x, r=2):
    b, c, d, h, wd = x.shape
    oc = c // r ** 3
    return x.view(b, oc, r, r, r, d, h, wd).permute(0, 1, 5, 2, 6, 3, 7, 4).reshape(b, oc, d * r, h * r, wd * r)


def decoder(q, w, tape=None):
    ys = _rec(tape, "d.ys", F.conv3d(q, w["decoder.stem.0.weight"], w["decoder.stem.0.bias"], padding=1))
    a = _rec(tape, "d.d2", _gn_relu(ys, w, "decoder.stem.1", 8))
    a = _rec(tape, "d.x6", _res_block(a, w, "decoder.res_stack.0", tape=tape, tag="d.r64"))
    a = _rec(tape, "d.x7", _attention(a, w, "decoder.attn"))
    up = _rec(tape, "d.up", F.conv3d(a, w["decoder.up_conv.weight"], w["decoder.up_conv.bias"], padding=1))
    pre = _rec(tape, "d.pre", F.conv3d(pixel_shuffle3d(up), w["decoder.final.weight"], w["decoder.final.bias"], padding=1))
    return torch.sigmoid(pre)


def quantize(z, codebook, commitment_cost=0.25):
    """Assignment with the expanded distance, straight-through output, commitment loss (VQVAE_v2.py:107-150); no EMA here."""
    flat = z.permute(0, 2, 3, 4, 1).reshape(-1, z.shape[1])
    d = (flat ** 2).sum(1, keepdim=True) + (codebook ** 2).sum(1) - 2 * flat @ codebook.t()
    idx = d.argmin(1)
    q = codebook[idx].view(z.shape[0], *z.shape[2:], z.shape[1]).permute(0, 4, 1, 2, 3)
    loss = commitment_cost * F.mse_loss(z, q.detach())
    return z + (q - z).detach(), loss, idx


def training_loss(x, w, commitment_cost=0.25, mse_weight=0.8, l1_weight=0.2, tape=None):
    """Forward of one training step: returns (loss, dict of pieces).  x: [B,1,8,8,8].  tape: dict receiving intermediates."""
    z = _rec(tape, "z", encoder(x, w, tape))
    q, vq_loss, idx = quantize(z, w["quantizer.embedding"], commitment_cost)
    q = _rec(tape, "q", q)
    recon = decoder(q, w, tape)
    mse, l1 = F.mse_loss(recon, x), F.l1_loss(recon, x)
    loss = mse_weight * mse + l1_weight * l1 + vq_loss
    return loss, {"z": z, "idx": idx, "recon": recon, "mse": mse, "l1": l1, "vq_loss": vq_loss}


def grads(x, weights: dict, **kw):
    """weights: name -> numpy/tensor.  Returns (loss float, pieces, {name: grad tensor}) for the trainable tensors."""
    w = {k: torch.as_tensor(v, dtype=torch.float32).clone().requires_grad_(not k.startswith("quantizer.")) for k, v in weights.items()}
    loss, pieces = training_loss(torch.as_tensor(x, dtype=torch.float32).view(-1, 1, 8, 8, 8), w, **kw)
    loss.backward()
    return float(loss.detach()), pieces, {k: v.grad for k, v in w.items() if v.requires_grad}


def adamw_step(params: dict, grads_: dict, state: dict, lr: float, step: int, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-4):
    """torch.optim.AdamW's update (decoupled weight decay), functional; state: name -> (m, v)."""
    b1, b2 = betas
    for k, p in params.items():
        g = grads_[k]
        m, v = state.setdefault(k, (torch.zeros_like(p), torch.zeros_like(p)))
        p.mul_(1 - lr * weight_decay)
        m.mul_(b1).add_(g, alpha=1 - b1)
        v.mul_(b2).addcmul_(g, g, value=1 - b2)
        denom = (v.sqrt() / (1 - b2 ** step) ** 0.5).add_(eps)
        p.addcdiv_(m, denom, value=-lr / (1 - b1 ** step))
