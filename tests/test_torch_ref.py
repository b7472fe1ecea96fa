"""The PyTorch fp32 restatement used as gradient oracle (tests/torch_ref.py) against golden loss / gradient summaries of the
imported reference model (tests/golden/make_golden_grads.py)."""
import os

import numpy as np
import torch

import torch_ref
from vqvdb_amd import synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_restatement_matches_reference_loss_and_gradients(weights):
    g = np.load(os.path.join(ROOT, "tests", "golden", "golden_grads_v1.npz"))
    torch.set_num_threads(8)
    loss, pieces, grads = torch_ref.grads(synth.make_leaves(16, seed=5000), weights)
    assert abs(loss - float(g["loss"])) < 1e-6 * abs(float(g["loss"]))
    for k in ("mse", "l1", "vq_loss"):
        assert abs(float(pieces[k]) - float(g[k])) < 1e-5 * abs(float(g[k])), k
    names = [k[2:] for k in g.files if k.startswith("g:")]
    assert sorted(names) == sorted(grads)
    for name in names:
        want = g["g:" + name]
        got = grads[name].double().flatten()
        scale = max(want[1], 1e-12)                          # L2 norm of the reference gradient
        assert abs(got.norm().item() - want[1]) < 1e-4 * scale, name
        assert abs(got.sum().item() - want[0]) < 1e-4 * scale * max(1.0, got.numel() ** 0.5), name
        assert np.abs(got[:6].numpy() - want[2:]).max() < 1e-4 * scale, name


def test_restatement_loop_matches_the_reference_optimizer_and_schedule(weights):
    """Five steps of the training loop against tests/golden/golden_trainloop_v1.npz — what the IMPORTED reference produced with
    torch.optim.AdamW + CosineAnnealingLR (python/training.py:98-101,136-164): pins torch_ref.adamw_step, full_training.cosine_lr and
    the EMA formula the GPU tests use as their loop oracle to the reference's own optimizer, not to a builder-written one."""
    from vqvdb_amd.full_training import cosine_lr
    fx = np.load(os.path.join(ROOT, "tests", "golden", "golden_trainloop_v1.npz"))
    torch.set_num_threads(8)
    w = {k: torch.as_tensor(v).clone() for k, v in weights.items()}
    params = {k: v for k, v in w.items() if not k.startswith("quantizer.")}
    cs, avg, state = torch.ones(256), w["quantizer.embedding"].clone(), {}
    t_max = int(fx["t_max"])
    for s, seed in enumerate(fx["seeds"].tolist()):
        x = torch.as_tensor(synth.make_leaves(64, seed=int(seed))).view(-1, 1, 8, 8, 8)
        for p in params.values():
            p.requires_grad_(True)
            p.grad = None
        loss, pieces = torch_ref.training_loss(x, w)
        loss.backward()
        lr = cosine_lr(1e-4, s, t_max)
        assert abs(lr - float(fx[f"s{s}/lr"])) < 1e-12
        for k, got in (("loss", loss), ("mse", pieces["mse"]), ("l1", pieces["l1"]), ("vq_loss", pieces["vq_loss"])):
            assert abs(float(got) - float(fx[f"s{s}/{k}"])) < 2e-5 * abs(float(fx[f"s{s}/{k}"])), (s, k)
        with torch.no_grad():
            if s == 0:
                for name in [k[4:] for k in fx.files if k.startswith("g32/")]:
                    g, ref = params[name].grad.numpy(), fx["g32/" + name]
                    assert np.abs(g - ref).max() < 2e-5 * np.abs(ref).max(), name
            flat = pieces["z"].detach().permute(0, 2, 3, 4, 1).reshape(-1, 128)
            enc = torch.nn.functional.one_hot(pieces["idx"], 256).float()
            probs = enc.mean(0)
            ppl = float(torch.exp(-(probs * torch.log(probs + 1e-10)).sum()))
            assert abs(ppl - float(fx[f"s{s}/perplexity"])) < 1e-4 * float(fx[f"s{s}/perplexity"])
            cs = cs * 0.95 + (1 - 0.95) * enc.sum(0)
            avg = avg * 0.95 + (1 - 0.95) * (enc.t() @ flat)
            w["quantizer.embedding"] = avg / cs.clamp(min=1e-4)[:, None]
            grads = {k: p.grad for k, p in params.items()}
            for p in params.values():
                p.requires_grad_(False)
            torch_ref.adamw_step(params, grads, state, lr=lr, step=s + 1)
        assert np.abs(cs.numpy() - fx[f"s{s}/cluster_size"]).max() < 1e-5
        assert np.abs(w["quantizer.embedding"].numpy() - fx[f"s{s}/embedding"]).max() < 1e-4 * np.abs(fx[f"s{s}/embedding"]).max()
    lr0 = 1e-4
    for name in [k[3:] for k in fx.files if k.startswith("p5/")]:
        diff = np.abs(params[name].numpy() - fx["p5/" + name])
        # Adam's first steps move every element by ~lr whatever the gradient's size: an element whose gradient is zero within rounding
        # may differ by up to 2 lr per step; the bulk must agree far better
        assert diff.max() <= 10.5 * lr0 and diff.mean() < 0.05 * lr0, (name, float(diff.max()), float(diff.mean()))
    for name in [k[5:] for k in fx.files if k.startswith("psum/")]:
        v = params[name].double()
        assert abs(float(v.norm()) - fx["psum/" + name][1]) < 1e-4 * max(fx["psum/" + name][1], 1e-3), name
