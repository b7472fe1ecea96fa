#!/usr/bin/env python3
"""Wide golden net: outputs of the IMPORTED reference model on enough positions to see ppm effects.

Runs only in the build container (needs /root/reference and CPU torch).  Inputs are regenerated at test time from
vqvdb_amd.synth (their SHA-256 is stored so generator drift is caught); only the reference's OUTPUTS are stored.

    python tests/golden/make_golden_wide.py        ->  tests/golden/golden_wide_v1.npz

Pinned (reference: python/VQVAE_v2.py:350-377):
  * idx_uniform  u8 [16384,64]  VQVAE.encode on synth.make_leaves(16384, seed=4321)            (1 048 576 positions)
  * idx_sparse   u8 [4096,64]   VQVAE.encode on synth.sparse_leaves(4096, seed=2468)            (background-dominated content)
  * idx_edge     u8 [8,64]      VQVAE.encode on synth.edge_leaves()
  * tie_pos / tie_gap: every flat position (uniform | sparse | edge, concatenated) whose relative top-2 distance gap is
    < 1e-4, with the gap — a position that is NOT listed has a gap >= 1e-4 and must match bit-exactly
  * rec          f32 [2048,512] VQVAE.decode of idx_uniform[:1024] | idx_sparse[:1024]
  * act_<layer>  per-layer activations (forward hooks) of uniform leaves 0, 1 and sparse leaves 0..2 — 5 leaves
  * batch independence asserted here: encode at B = 1, 63, 65 agrees with the big-batch run
"""
import hashlib
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)
sys.path.insert(0, "/root/reference/python")

from vqvdb_amd import synth  # noqa: E402
from make_golden import build_model, decode, encode_with_gap  # noqa: E402  (same reference import as golden_v1)

torch.set_num_threads(8)
TIE_THR = 1e-4
LAYERS = lambda m: [  # noqa: E731
    ("enc_pre0", m.encoder.pre[0]), ("enc_pre2", m.encoder.pre[2]), ("enc_pre3", m.encoder.pre[3]), ("enc_down", m.encoder.down),
    ("enc_res", m.encoder.res_stack), ("enc_attn", m.encoder.attn), ("enc_proj", m.encoder.proj),
    ("dec_stem0", m.decoder.stem[0]), ("dec_stem", m.decoder.stem), ("dec_res", m.decoder.res_stack), ("dec_attn", m.decoder.attn),
    ("dec_up", m.decoder.up_conv), ("dec_ps", m.decoder.pixshuf), ("dec_final", m.decoder.final)]


def sha(a: np.ndarray) -> str:
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def inputs():
    return synth.make_leaves(16384, seed=4321), synth.sparse_leaves(4096, seed=2468), synth.edge_leaves()


def main(out_path=None):
    m = build_model()
    uni, spa, edge = inputs()
    idx, gaps = [], []
    for part in (uni, spa, edge):
        ii, gg = [], []
        for s in range(0, len(part), 2048):                      # bounded memory: the [B*64, 256] distance matrix
            i, g, _ = encode_with_gap(m, part[s:s + 2048])
            ii.append(i), gg.append(g)
        idx.append(np.concatenate(ii)), gaps.append(np.concatenate(gg).reshape(-1))
    for b in (1, 63, 65):                                        # batch independence
        assert np.array_equal(encode_with_gap(m, uni[:b])[0], idx[0][:b]), b
        assert np.array_equal(encode_with_gap(m, spa[100:100 + b])[0], idx[1][100:100 + b]), b
    flat_gap = np.concatenate(gaps)
    tie_pos = np.nonzero(flat_gap < TIE_THR)[0].astype(np.int64)
    dec_idx = np.concatenate([idx[0][:1024], idx[1][:1024]])
    rec = np.concatenate([decode(m, dec_idx[s:s + 256]) for s in range(0, 2048, 256)])

    acts = {}
    act_leaves = np.concatenate([uni[:2], spa[:3]])
    cur = []

    def hook(name):
        def f(_mod, _inp, outp):
            cur.append((name, outp.detach().numpy().reshape(outp.shape[1], -1).copy()))
        return f
    hs = [mod.register_forward_hook(hook(name)) for name, mod in LAYERS(m)]
    with torch.no_grad():
        for li in range(len(act_leaves)):
            cur.clear()
            x0 = torch.from_numpy(act_leaves[li:li + 1]).view(1, 1, 8, 8, 8)
            m.decode(m.encode(x0))
            for name, a in cur:
                acts.setdefault(name, []).append(a)
    for h in hs:
        h.remove()

    out = dict(idx_uniform=idx[0], idx_sparse=idx[1], idx_edge=idx[2], tie_pos=tie_pos, tie_gap=flat_gap[tie_pos].astype(np.float32),
               tie_thr=np.float32(TIE_THR), rec=rec.astype(np.float32),
               sha_uniform=np.array(sha(uni)), sha_sparse=np.array(sha(spa)), sha_edge=np.array(sha(edge)))
    for k, v in acts.items():
        out["act_" + k] = np.stack(v).astype(np.float32)
    path = out_path or os.path.join(HERE, "golden_wide_v1.npz")
    np.savez_compressed(path, **out)
    print(f"positions {flat_gap.size}; near-ties (<{TIE_THR}): {len(tie_pos)}; min gap {flat_gap.min():.3e}; gaps < 1e-5: {(flat_gap < 1e-5).sum()}")
    print(f"codes used: uniform {len(np.unique(idx[0]))}, sparse {len(np.unique(idx[1]))}; file {os.path.getsize(path) / 1e6:.2f} MB")


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else None)
