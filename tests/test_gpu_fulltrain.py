"""GPU tests of the full training step (stage 2): HIP forward / backward against PyTorch autograd of the plain fp32
restatement in tests/torch_ref.py (itself pinned to the imported reference by tests/test_torch_ref.py).  Tolerances are
relative to each tensor's max magnitude."""
import numpy as np
import pytest
import torch

import torch_ref
from vqvdb_amd import synth, weightpack
from vqvdb_amd.codec import HipCodec

pytestmark = pytest.mark.gpu
N = 40   # two tiles, the second one ragged
TOL_GRAD = 5e-4   # parameter gradients are long fp32 sums with cancellation on both sides (HIP and torch)


def _rel(a, b):
    return float(np.abs(np.asarray(a, np.float64) - np.asarray(b, np.float64)).max() / max(float(np.abs(b).max()), 1e-30))


@pytest.fixture(scope="module")
def ref(weights):
    torch.set_num_threads(16)
    x = synth.make_leaves(N, seed=6000)
    w = {k: torch.as_tensor(v).clone().requires_grad_(not k.startswith("quantizer.")) for k, v in weights.items()}
    keep = {}
    loss, pieces = torch_ref.training_loss(torch.as_tensor(x).view(-1, 1, 8, 8, 8), w)
    return {"x": x, "w": w, "loss": loss, "pieces": pieces, "keep": keep}


@pytest.fixture()
def fcodec(weights):
    c = HipCodec(weightpack.dumps(weights))
    c.fulltrain_begin()
    yield c
    c.close()


def test_training_forward_matches_torch(fcodec, ref, weights):
    assert fcodec.fulltrain_param_count() == sum(v.size for k, v in weights.items() if not k.startswith("quantizer.")) == 995905
    x = torch.from_numpy(ref["x"]).cuda()
    fcodec.fulltrain_forward_device(x.data_ptr(), N)
    torch.cuda.synchronize()
    p = ref["pieces"]
    z = p["z"].detach().numpy().reshape(N, 128, 64)
    recon = p["recon"].detach().numpy().reshape(N, 1, 512)
    assert _rel(fcodec.fetch("t_recon", N, 1, 512), recon) < 1e-5
    q = fcodec.fetch("t_q", N, 128, 64)
    E = weights["quantizer.embedding"]
    idx = p["idx"].numpy().reshape(N, 64)
    assert np.array_equal(q, E[idx].transpose(0, 2, 1))


@pytest.fixture(scope="module")
def ref_grads(weights):
    torch.set_num_threads(16)
    x = synth.make_leaves(N, seed=6000)
    w = {k: torch.as_tensor(v).clone().requires_grad_(not k.startswith("quantizer.")) for k, v in weights.items()}
    tape = {}
    loss, pieces = torch_ref.training_loss(torch.as_tensor(x).view(-1, 1, 8, 8, 8), w, tape=tape)
    loss.backward()
    return {"x": x, "w": w, "tape": tape, "loss": float(loss.detach())}


def _hip_grads(fcodec, weights, x):
    xs = torch.from_numpy(x).cuda()
    G = torch.zeros(fcodec.fulltrain_param_count(), device="cuda")
    fcodec.fulltrain_fwdbwd_device(xs.data_ptr(), len(x), len(x), G.data_ptr())
    torch.cuda.synchronize()
    flat = G.cpu().numpy()
    out, off = {}, 0
    for name, shape, _ in synth.TENSORS:
        if name.startswith("quantizer."):
            continue
        n = int(np.prod(shape))
        out[name] = flat[off:off + n].reshape(shape)
        off += n
    assert off == len(flat)
    return out


def test_decoder_gradients_match_autograd(fcodec, ref_grads, weights):
    got = _hip_grads(fcodec, weights, ref_grads["x"])
    tape = ref_grads["tape"]
    assert _rel(fcodec.fetch("g_pre", N, 1, 512), tape["d.pre"].grad.numpy().reshape(N, 1, 512)) < 1e-5
    up = tape["d.up"].grad.numpy().reshape(N, 256, 64)
    assert _rel(fcodec.fetch("g_upA", N, 128, 64), up[:, :128]) < 1e-5 and _rel(fcodec.fetch("g_upB", N, 128, 64), up[:, 128:]) < 1e-5
    bad = [(name, _rel(g, ref_grads["w"][name].grad.numpy())) for name, g in got.items() if name.startswith("decoder.")]
    assert all(e < TOL_GRAD for _, e in bad), [b for b in bad if b[1] >= TOL_GRAD]


def test_encoder_gradients_match_autograd(fcodec, ref_grads, weights):
    got = _hip_grads(fcodec, weights, ref_grads["x"])
    tape = ref_grads["tape"]
    assert _rel(fcodec.fetch("g_q", N, 128, 64), tape["z"].grad.numpy().reshape(N, 128, 64)) < 1e-4        # after the straight-through step: dz
    assert _rel(fcodec.fetch("g32c", N, 32, 64), tape["e.x7"].grad.numpy().reshape(N, 32, 64)) < 1e-4
    assert _rel(fcodec.fetch("g16a", N, 16, 512), tape["e.a6"].grad.numpy().reshape(N, 16, 512)) < 1e-4
    assert _rel(fcodec.fetch("g16b", N, 16, 512), tape["e.y1"].grad.numpy().reshape(N, 16, 512)) < 1e-4
    bad = [(name, _rel(g, ref_grads["w"][name].grad.numpy())) for name, g in got.items() if name.startswith("encoder.")]
    assert all(e < TOL_GRAD for _, e in bad), [b for b in bad if b[1] >= TOL_GRAD]
