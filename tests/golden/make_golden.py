#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ by IMPORTING the reference model.

Runs only in the build container (needs /root/reference and CPU torch); the
outputs are plain data (inputs are regenerated from vqvdb_amd.synth, so only the
reference's OUTPUTS are stored).  No reference source is copied.

    python tests/golden/make_golden.py

What is pinned:
  * idx_rand   u8 [1024,64]  VQVAE.encode (python/VQVAE_v2.py:350-369) on synth.make_leaves(1024,1234)
  * idx_edge   u8 [8,64]     same on synth.edge_leaves()
  * near-tie list: flat positions whose relative top-2 distance gap < 1e-3, with the gap
  * rec_rand   f32 [64,512]  VQVAE.decode (VQVAE_v2.py:371-377) of idx_rand[:64]
  * rec_edge   f32 [8,512]   decode of idx_edge
  * act_*      per-layer activations of random leaf 0 (forward hooks), NCDHW flattened [C, D*H*W]
  * batch-independence: encode(B=1) and encode(B=65) agree with the B=1024 run (asserted here)
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, "/root/reference/python")

from vqvdb_amd import synth  # noqa: E402
from VQVAE_v2 import VQVAE  # noqa: E402  (the reference model, imported, not copied)

torch.set_num_threads(8)
torch.manual_seed(0)


def build_model():
    w = synth.make_weights(seed=0)
    m = VQVAE(1, synth.D_EMBED, synth.K_CODES, 0.25).eval()
    sd = {k: torch.from_numpy(v) for k, v in w.items()}
    sd["quantizer.cluster_size"] = torch.ones(synth.K_CODES)
    sd["quantizer.embed_avg"] = sd["quantizer.embedding"].clone()
    missing = m.load_state_dict(sd, strict=True)
    assert not missing.missing_keys and not missing.unexpected_keys
    return m


@torch.no_grad()
def encode_with_gap(m, leaves):
    """Reference encode + the relative top-2 gap per position (same formula as
    VQVAE.encode, VQVAE_v2.py:358-367)."""
    x = torch.from_numpy(leaves).view(-1, 1, 8, 8, 8)
    idx = m.encode(x)                                   # [B,4,4,4] int64
    z = m.encoder(x)
    flat = z.permute(0, 2, 3, 4, 1).contiguous().view(-1, synth.D_EMBED)
    E = m.quantizer.embedding
    d = (flat ** 2).sum(1, keepdim=True) + (E ** 2).sum(1) - 2 * flat @ E.t()
    d2, _ = torch.topk(d, 2, dim=1, largest=False)
    gap = ((d2[:, 1] - d2[:, 0]) / d2[:, 1].abs().clamp_min(1e-30)).numpy()
    assert torch.equal(idx.view(-1), d.argmin(1))
    return idx.view(-1, 64).numpy().astype(np.uint8), gap.reshape(-1, 64).astype(np.float32), z.numpy()


@torch.no_grad()
def decode(m, idx_u8):
    idx = torch.from_numpy(idx_u8.astype(np.int64)).view(-1, 4, 4, 4)
    return m.decode(idx).view(-1, 512).numpy()


def main():
    m = build_model()
    rand = synth.make_leaves(1024, seed=1234)
    edge = synth.edge_leaves()

    idx_rand, gap_rand, z_rand = encode_with_gap(m, rand)
    idx_edge, gap_edge, _ = encode_with_gap(m, edge)

    # batch independence
    i1, _, _ = encode_with_gap(m, rand[:1])
    i65, _, _ = encode_with_gap(m, rand[:65])
    assert np.array_equal(i1, idx_rand[:1]) and np.array_equal(i65, idx_rand[:65])

    rec_rand = decode(m, idx_rand[:64])
    rec_edge = decode(m, idx_edge)

    # per-layer activations of random leaf 0 via forward hooks
    acts = {}

    def hook(name):
        def f(_mod, _inp, out):
            acts[name] = out.detach().numpy().reshape(out.shape[1], -1).copy()
        return f

    hs = []
    for name, mod in [
        ("enc_pre0", m.encoder.pre[0]), ("enc_pre2", m.encoder.pre[2]), ("enc_pre3", m.encoder.pre[3]),
        ("enc_down", m.encoder.down), ("enc_res", m.encoder.res_stack), ("enc_attn", m.encoder.attn),
        ("enc_proj", m.encoder.proj),
        ("dec_stem0", m.decoder.stem[0]), ("dec_stem", m.decoder.stem), ("dec_res", m.decoder.res_stack),
        ("dec_attn", m.decoder.attn), ("dec_up", m.decoder.up_conv), ("dec_ps", m.decoder.pixshuf),
        ("dec_final", m.decoder.final),
    ]:
        hs.append(mod.register_forward_hook(hook(name)))
    with torch.no_grad():
        x0 = torch.from_numpy(rand[:1]).view(1, 1, 8, 8, 8)
        i0 = m.encode(x0)
        m.decode(i0)
    for h in hs:
        h.remove()

    tie_thr = 1e-3
    flat_gap = np.concatenate([gap_rand.reshape(-1), gap_edge.reshape(-1)])
    tie_pos = np.nonzero(flat_gap < tie_thr)[0].astype(np.int32)
    out = dict(
        idx_rand=idx_rand, idx_edge=idx_edge,
        tie_pos=tie_pos, tie_gap=flat_gap[tie_pos].astype(np.float32), tie_thr=np.float32(tie_thr),
        rec_rand=rec_rand.astype(np.float32), rec_edge=rec_edge.astype(np.float32),
        z_rand0=z_rand[0].reshape(128, 64).astype(np.float32),
    )
    for k, v in acts.items():
        out["act_" + k] = v.astype(np.float32)
    np.savez_compressed(os.path.join(HERE, "golden_v1.npz"), **out)

    used = len(np.unique(idx_rand))
    print(f"codes used by 1024 random leaves: {used}/256")
    print(f"near-ties (<{tie_thr}): {len(tie_pos)} of {flat_gap.size}; min gap {flat_gap.min():.3e}")
    print(f"idx_rand sum {int(idx_rand.astype(np.int64).sum())}  rec_rand sum {float(rec_rand.astype(np.float64).sum()):.6f}")
    print(f"z std {z_rand.std():.4f}  codebook std {m.quantizer.embedding.std():.4f}")
    print({k: v.shape for k, v in out.items() if hasattr(v, 'shape')})


if __name__ == "__main__":
    main()
