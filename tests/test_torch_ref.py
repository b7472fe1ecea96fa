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
