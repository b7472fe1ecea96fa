#!/usr/bin/env python3
"""Golden vectors on FIVE MORE WEIGHT REGIMES: outputs of the IMPORTED reference model with weights
other than synth.make_weights(0), which every other fixture uses.

Runs only in the build container (needs /root/reference and CPU torch).  Nothing of the reference is copied: the file holds
reference OUTPUTS plus two state dicts the reference itself produced (data, not source).

    python tests/golden/make_golden_regimes.py [out.npz]     ->  tests/golden/golden_regimes_v1.npz

Regimes (the create-time algebra of the HIP backend — Ep = E.P and c_k, the (tap, code) stem table, the folded tail — is
weight-dependent, so each regime is a different population of near-ties and magnitudes):
  * seed1, seed2   synth.make_weights(1) / (2): outputs only, the weights regenerate on any box.
  * trained        seed-0 weights loaded into the imported VQVAE, then the reference's own optimisation step
                   (python/training.py:136-164: loss 0.8.MSE + 0.2.L1 + vq_loss, AdamW lr 1e-4 wd 1e-4 betas (0.9, 0.999),
                   CosineAnnealingLR over the run) in fp32 with model.train(), so the EMA codebook moves every step
                   (python/VQVAE_v2.py:107-156); batches of 256 leaves = half synth.sparse_leaves, half synth.make_leaves;
                   one VQVAE.check_and_reset_dead_codes (:382-417) after step RESET_AT with the encoder outputs of that
                   step's batch.  The resulting state_dict is stored (w_trained/<tensor name>), cluster_size included, so the
                   test can see which codes are dead (dead codes decay towards the origin once cluster_size < eps, :146-147).
  * deadcodes      the trained encoder / decoder with the three quantizer buffers as they stood right BEFORE the reset: never-chosen
                   codes have cluster_size < eps and have been shrinking towards the origin (w_deadcodes/quantizer.*; the other
                   tensors are w_trained's).
  * default        torch.manual_seed(0); VQVAE(1, 128, 256, 0.25): the reference's default init (residual conv2 ~ N(0, 1e-3²),
                   GroupNorm affine = identity, unit-norm codebook rows: ‖e‖² identical for every code, :99-101, :201-202).
                   state_dict stored (w_default/<tensor name>).
Per regime <r>:
  * <r>/idx_uniform u8 [4096,64]  VQVAE.encode on synth.make_leaves(4096, seed=9001)
  * <r>/idx_sparse  u8 [2048,64]  VQVAE.encode on synth.sparse_leaves(2048, seed=9002)
  * <r>/idx_edge    u8 [8,64]     VQVAE.encode on synth.edge_leaves()
  * <r>/tie_pos, <r>/tie_gap      every flat position (uniform | sparse | edge) with a relative top-2 gap < 1e-4, and the gap
  * <r>/rec         f32 [512,512] VQVAE.decode of idx_uniform[:256] | idx_sparse[:256]
  * <r>/rec64       f64 [512,512] the same decode by the reference model converted to fp64 (.double()), stored for the regimes whose
                    outputs saturate (min < 1e-3: trained, deadcodes); <r>/ref32_vs_ref64 = the largest element-wise relative
                    distance between the two evaluations (the reference's own fp32 rounding), every regime
  * <r>/act_<layer> activations (forward hooks) of uniform leaf 0 and sparse leaf 0
"""
import copy
import hashlib
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
from VQVAE_v2 import VQVAE  # noqa: E402  (the reference model, imported, not copied)
from make_golden import decode, encode_with_gap  # noqa: E402
from make_golden_wide import LAYERS  # noqa: E402

torch.set_num_threads(8)
TIE_THR = 1e-4
TRAIN_STEPS = 360
TRAIN_BATCH = 256
RESET_AT = 240
N_UNI, N_SPA = 4096, 2048
SEED_UNI, SEED_SPA = 9001, 9002


def sha(a: np.ndarray) -> str:
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def model_from(w: dict) -> VQVAE:
    m = VQVAE(1, synth.D_EMBED, synth.K_CODES, 0.25)
    sd = {k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in w.items()}
    sd.setdefault("quantizer.cluster_size", torch.ones(synth.K_CODES))
    sd.setdefault("quantizer.embed_avg", sd["quantizer.embedding"].clone())
    r = m.load_state_dict(sd, strict=True)
    assert not r.missing_keys and not r.unexpected_keys
    return m


def train_reference(m: VQVAE):
    """The reference's step (training.py:136-164) without autocast/GradScaler (fp32), on synthetic leaves."""
    opt = torch.optim.AdamW(m.parameters(), lr=1e-4, weight_decay=1e-4, betas=(0.9, 0.999))          # training.py:98
    sched = torch.optim.lr_scheduler.CosineAnnealingLR(opt, T_max=TRAIN_STEPS)                      # training.py:101
    half = TRAIN_BATCH // 2
    spa = synth.sparse_leaves(half * TRAIN_STEPS, seed=5150)
    log = []
    m.train()
    for step in range(TRAIN_STEPS):
        x = np.concatenate([spa[step * half:(step + 1) * half], synth.make_leaves(half, seed=6000 + step)])
        x = torch.from_numpy(x).view(-1, 1, 8, 8, 8)
        opt.zero_grad()
        z, recon, vq_loss, ppl = m(x)
        loss = 0.8 * F.mse_loss(recon, x) + 0.2 * F.l1_loss(recon, x) + vq_loss
        loss.backward()
        opt.step()
        sched.step()
        if step == RESET_AT:
            dead = int((m.quantizer.cluster_size < 1.0).sum())
            pre_reset = {k: getattr(m.quantizer, k).detach().numpy().copy() for k in ("embedding", "cluster_size", "embed_avg")}
            torch.manual_seed(4242)                                  # check_and_reset_dead_codes draws with torch.randint (:404)
            m.check_and_reset_dead_codes(z.detach())
            log.append(f"step {step}: reset {dead} dead codes")
        if step % 40 == 0 or step == TRAIN_STEPS - 1:
            log.append(f"step {step}: loss {loss.item():.5f} vq {vq_loss.item():.5f} perplexity {ppl.item():.1f}")
    m.eval()
    return log, pre_reset


def regime_outputs(m: VQVAE, prefix: str, out: dict):
    m.eval()
    uni, spa, edge = synth.make_leaves(N_UNI, seed=SEED_UNI), synth.sparse_leaves(N_SPA, seed=SEED_SPA), synth.edge_leaves()
    idx, gaps = [], []
    for part in (uni, spa, edge):
        ii, gg = [], []
        for s in range(0, len(part), 2048):
            i, g, _ = encode_with_gap(m, part[s:s + 2048])
            ii.append(i), gg.append(g)
        idx.append(np.concatenate(ii)), gaps.append(np.concatenate(gg).reshape(-1))
    for b in (1, 65):                                                # batch independence
        assert np.array_equal(encode_with_gap(m, uni[:b])[0], idx[0][:b]), b
    flat_gap = np.concatenate(gaps)
    tie_pos = np.nonzero(flat_gap < TIE_THR)[0].astype(np.int64)
    dec_idx = np.concatenate([idx[0][:256], idx[1][:256]])
    rec = np.concatenate([decode(m, dec_idx[s:s + 256]) for s in range(0, 512, 256)])
    assert np.isfinite(rec).all()
    # the same decode with the reference model in fp64 (same weights, .double()): where the sigmoid saturates (a trained model's
    # background voxels, outputs down to 1e-9) the element-wise relative error of ANY fp32 evaluation is the absolute error of a
    # pre-activation of magnitude ~20, and the reference's own fp32 run is 1.5e-5 away from this
    m64 = copy.deepcopy(m).double().eval()
    with torch.no_grad():
        rec64 = m64.decode(torch.from_numpy(dec_idx.astype(np.int64)).view(-1, 4, 4, 4)).view(-1, 512).numpy()
    acts, cur = {}, []

    def hook(name):
        def f(_mod, _inp, outp):
            cur.append((name, outp.detach().numpy().reshape(outp.shape[1], -1).copy()))
        return f
    hs = [mod.register_forward_hook(hook(name)) for name, mod in LAYERS(m)]
    with torch.no_grad():
        for leaf in (uni[0], spa[0]):
            cur.clear()
            m.decode(m.encode(torch.from_numpy(leaf).view(1, 1, 8, 8, 8)))
            for name, a in cur:
                acts.setdefault(name, []).append(a)
    for h in hs:
        h.remove()
    out[f"{prefix}/idx_uniform"], out[f"{prefix}/idx_sparse"], out[f"{prefix}/idx_edge"] = idx
    out[f"{prefix}/tie_pos"] = tie_pos
    out[f"{prefix}/tie_gap"] = flat_gap[tie_pos].astype(np.float32)
    out[f"{prefix}/rec"] = rec.astype(np.float32)
    if rec.min() < 1e-3:      # saturating regimes only (4 MB each): elsewhere the fp32 outputs are compared directly at 1e-5
        out[f"{prefix}/rec64"] = rec64.astype(np.float64)
    out[f"{prefix}/ref32_vs_ref64"] = np.float64((np.abs(rec.astype(np.float64) - rec64) / rec64).max())
    for k, v in acts.items():
        out[f"{prefix}/act_{k}"] = np.stack(v).astype(np.float32)
    E = m.quantizer.embedding.numpy()
    print(f"[{prefix}] positions {flat_gap.size}; gaps < 1e-4: {len(tie_pos)}, < 1e-5: {(flat_gap < 1e-5).sum()}, min {flat_gap.min():.2e}; "
          f"codes used uniform {len(np.unique(idx[0]))} / sparse {len(np.unique(idx[1]))}; |e| min {np.linalg.norm(E, axis=1).min():.3e} "
          f"max {np.linalg.norm(E, axis=1).max():.3e}; decode: min output {rec.min():.2e}, ref32 vs ref64 {float(out[prefix + '/ref32_vs_ref64']):.2e}")


def main(out_path=None):
    out = {"tie_thr": np.float32(TIE_THR)}
    out["sha_uniform"] = np.array(sha(synth.make_leaves(N_UNI, seed=SEED_UNI)))
    out["sha_sparse"] = np.array(sha(synth.sparse_leaves(N_SPA, seed=SEED_SPA)))
    for seed in (1, 2):
        regime_outputs(model_from(synth.make_weights(seed)), f"seed{seed}", out)

    m = model_from(synth.make_weights(0))
    log, pre_reset = train_reference(m)
    for line in log:
        print("[trained]", line)
    sd = {k: v.detach().numpy().astype(np.float32).copy() for k, v in m.state_dict().items()}
    for k, v in sd.items():
        out[f"w_trained/{k}"] = v
    w0 = synth.make_weights(0)
    moved = {k: float(np.abs(sd[k] - w0[k]).max() / max(np.abs(w0[k]).max(), 1e-30)) for k in w0}
    print("[trained] largest relative moves:", sorted(moved.items(), key=lambda kv: -kv[1])[:4])
    print(f"[trained] cluster_size < 1: {(sd['quantizer.cluster_size'] < 1).sum()}, < 1e-4 (clamped): {(sd['quantizer.cluster_size'] < 1e-4).sum()}")
    regime_outputs(m, "trained", out)

    # deadcodes: the trained encoder / decoder with the codebook as it stood right BEFORE the dead-code reset — codes that were never
    # chosen have cluster_size < eps, so embedding = embed_avg / eps has been shrinking by 0.95 per step (VQVAE_v2.py:146-147): a
    # cluster of tiny-norm codes near the origin next to the live ones.  Only the three quantizer buffers are stored.
    with torch.no_grad():
        for k, v in pre_reset.items():
            getattr(m.quantizer, k).copy_(torch.from_numpy(v))
            out[f"w_deadcodes/quantizer.{k}"] = v.astype(np.float32)
    print(f"[deadcodes] cluster_size < 1: {(pre_reset['cluster_size'] < 1).sum()}, < 1e-4 (clamped): {(pre_reset['cluster_size'] < 1e-4).sum()}")
    regime_outputs(m, "deadcodes", out)

    torch.manual_seed(0)
    md = VQVAE(1, synth.D_EMBED, synth.K_CODES, 0.25).eval()
    for k, v in md.state_dict().items():
        out[f"w_default/{k}"] = v.detach().numpy().astype(np.float32).copy()
    regime_outputs(md, "default", out)

    path = out_path or os.path.join(HERE, "golden_regimes_v1.npz")
    np.savez_compressed(path, **out)
    print(f"file {os.path.getsize(path) / 1e6:.2f} MB, {len(out)} arrays")


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else None)
