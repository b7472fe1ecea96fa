#!/usr/bin/env python3
"""Epoch-level driver for codebook (EMA) training on the HIP backend — the quantizer part of the reference's
`train(args)` (python/training.py:47-258) and of BASELINE configs[4], one process per GPU:

    python -m vqvdb_amd.train_codebook train --pack model.vqw --model_path out/quantizer.npz [--data_dir DIR] ...
    python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 -m vqvdb_amd.train_codebook train ...

Same loop shape as the reference: 80/20 split (:77-81), per-epoch training pass over global batches of
`batch_size` leaves per rank, validation pass (:183-199), dead-code reset every 5 epochs from the first batch's
encoder outputs (:120,165-166,180-181), best-validation checkpoint (:216-233) and a final save (:252).  What is trained
is the codebook (EMA); encoder/decoder weights stay as loaded from the pack.  Data: `.npy` files of shape [N,8,8,8]
float32 like the reference's VDBLeafDataset (python/VQVAE_v2.py:21-66), or synthetic uniform leaves when no directory
is given.  Every rank holds its shard of each global batch in HBM; the only collective is the all-reduce of the
statistics buffer (RCCL).
"""
from __future__ import annotations

import argparse
import glob
import os
import sys
import time

import numpy as np
import torch
import torch.distributed as dist

from vqvdb_amd.codebook_training import CodebookTrainer
from vqvdb_amd.full_training import FullTrainer
from vqvdb_amd.codec import HipCodec
from vqvdb_amd.sharding import shard_range

DEAD_CODE_RESET_INTERVAL = 5     # training.py:120
SUBSAMPLE = 6                    # training.py:72-73: every 6th leaf of the dataset


def load_leaves(data_dir, synthetic_leaves: int, seed: int) -> np.ndarray:
    """All leaves as float32 [N,512] (the reference subsamples every 6th block of its .npy files)."""
    if data_dir:
        files = sorted(glob.glob(os.path.join(data_dir, "*.npy")))
        if not files:
            raise ValueError(f"No .npy files found in {data_dir}")
        arrs = []
        for f in files:
            a = np.load(f, mmap_mode="r")
            if a.shape[1:] != (8, 8, 8):
                raise ValueError(f"File {f}: invalid shape {a.shape}. Expected suffix (8, 8, 8)")
            arrs.append(np.asarray(a[::SUBSAMPLE], dtype=np.float32).reshape(-1, 512))
        return np.concatenate(arrs)
    from vqvdb_amd import synth
    base = synth.make_leaves(min(synthetic_leaves, 65536), seed=seed)
    reps = -(-synthetic_leaves // len(base))
    return np.tile(base, (reps, 1))[:synthetic_leaves] if reps > 1 else base


def split_train_val(n: int, seed: int):
    """80 % / 20 % random split (training.py:77-81), identical on every rank."""
    perm = np.random.default_rng(seed).permutation(n)
    n_train = int(0.8 * n)
    return perm[:n_train], perm[n_train:]


def train(args) -> dict:
    distributed = "RANK" in os.environ and int(os.environ.get("WORLD_SIZE", "1")) > 1
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if distributed and not dist.is_initialized():
        dist.init_process_group(args.backend, **({"device_id": torch.device("cuda", local)} if args.backend == "nccl" else {}))
    if args.single_gpu_rehearsal:
        local = 0
    device = torch.device("cuda", local)
    torch.cuda.set_device(device)
    log = (lambda *a: print(*a, flush=True)) if rank == 0 else (lambda *a: None)

    codec = HipCodec(args.pack, device_id=local)
    leaves = load_leaves(args.data_dir, args.leaves_per_epoch * 5 // 4, args.seed)
    tr_ids, va_ids = split_train_val(len(leaves), args.seed)
    full = args.mode == "full"
    if full:   # AdamW(lr, wd 1e-4, betas 0.9/0.999) + CosineAnnealingLR(T_max = epochs * steps) like training.py:104-108
        trainer = FullTrainer(codec, lr=args.lr, commitment_cost=args.commitment_cost, ema_decay=args.decay, ema_eps=args.eps,
                              t_max=args.epochs * max(len(tr_ids) // (args.batch_size * world), 1), device=str(device))
    else:
        trainer = CodebookTrainer(codec, commitment_cost=args.commitment_cost, decay=args.decay, eps=args.eps, device=str(device))
    log(f"Dataset: {len(leaves)} leaves, train {len(tr_ids)}, val {len(va_ids)}; {world} rank(s) x batch {args.batch_size}")
    # this rank's shard of every global batch, resident in HBM (2 KiB per leaf)
    gb = args.batch_size * world
    steps_per_epoch = len(tr_ids) // gb
    # every rank sees the same sizes, so every rank raises here together (a rank with an empty shard would otherwise fail alone
    # inside a step while its peers block in all_reduce)
    if steps_per_epoch < 1:
        raise SystemExit(f"training set of {len(tr_ids)} leaves is smaller than one global batch ({world} x {args.batch_size}); lower --batch-size")
    if len(va_ids) < world:
        raise SystemExit(f"validation set of {len(va_ids)} leaves cannot give each of the {world} ranks a leaf")

    def shard(ids, step):
        lo, hi = shard_range(gb, rank, world)
        return ids[step * gb + lo: step * gb + hi]

    d_all = torch.from_numpy(np.ascontiguousarray(leaves)).to(device)
    best_val, history = float("inf"), []
    start_epoch = 0
    if args.resume:   # continue from a checkpoint written below: weights, quantizer buffers and (full mode) AdamW moments + step count
        ck = dict(np.load(args.resume))
        start_epoch = int(ck.pop("epoch", 0))
        best_val = float(ck.pop("best_val_loss", best_val))
        trainer.load_checkpoint(ck) if full else trainer.load_state_dict(ck)
        log(f"Resumed from {args.resume} at epoch {start_epoch}")
    os.makedirs(os.path.dirname(os.path.abspath(args.model_path)) or ".", exist_ok=True)
    for epoch in range(start_epoch, args.epochs):
        order = np.random.default_rng(args.seed + 1 + epoch).permutation(tr_ids)        # shuffle=True (training.py:87-94)
        t0 = time.perf_counter()
        tot_vq, last = 0.0, None
        first_batch = None
        for step in range(steps_per_epoch):
            batch = d_all[torch.from_numpy(shard(order, step)).to(device)]
            want = (step % args.log_every == 0) or step == steps_per_epoch - 1
            if full:
                if step == 0:
                    first_batch = batch
                m = trainer.step(batch, want_metrics=want)
            else:
                m = trainer.step(batch, keep_latent=(step == 0), want_metrics=want)
            if m is not None:
                last = m
                tot_vq += m["vq_loss"]
        torch.cuda.synchronize(device)
        dt = time.perf_counter() - t0
        if (epoch + 1) % DEAD_CODE_RESET_INTERVAL == 0:
            n_dead = trainer.reset_dead_codes(first_batch) if full else trainer.reset_dead_codes()
            if n_dead:
                log(f"INFO: Resetting {n_dead} dead codes.")
        # validation (training.py:183-199): whole validation set in global batches, metrics averaged over batches
        val = {"recon_error": 0.0, "vq_loss": 0.0, "recon_mse": 0.0, "recon_l1": 0.0}
        n_val = max(len(va_ids) // gb, 1)
        for step in range(n_val):
            ids = shard(va_ids, step) if len(va_ids) >= gb else va_ids[rank::world]
            mv = trainer.evaluate(d_all[torch.from_numpy(ids).to(device)])
            for k in val:
                val[k] += mv[k] / n_val
        val_loss = val["recon_error"] + val["vq_loss"]
        rec = {"epoch": epoch + 1, "train_loss": last.get("loss"), "train_vq_loss": last["vq_loss"], "perplexity": last["perplexity"], "codes_used": last["codes_used"],
               "val_loss": val_loss, **{f"val_{k}": v for k, v in val.items()}, "leaves_per_s": steps_per_epoch * gb / dt, "epoch_s": dt}
        history.append(rec)
        log(f"Epoch {epoch + 1:02d}/{args.epochs} | Train VQ: {last['vq_loss']:.6f} | Val Loss: {val_loss:.6f} | Perplexity: {last['perplexity']:.2f} | "
            f"{rec['leaves_per_s'] / 1e6:.3f} M leaves/s ({dt:.2f} s/epoch)")
        if val_loss < best_val and rank == 0:
            best_val = val_loss
            # full mode: optimizer moments and step count included (training.py:216-226), see FullTrainer.checkpoint
            np.savez(args.model_path, epoch=epoch + 1, best_val_loss=best_val, **(trainer.checkpoint() if full else trainer.state_dict()))
            log(f"New best validation loss: {val_loss:.6f} - model saved.")
    trainer.finish()
    if rank == 0:
        root, ext = os.path.splitext(args.model_path)
        sd = trainer.state_dict()
        np.savez(root + "_final" + (ext or ".npz"), epoch=args.epochs, **sd)
        if full:   # the role of the reference's scripted-model export (training.py:254-258): an inference artefact for this backend
            from vqvdb_amd import weightpack
            weightpack.save(root + "_final.vqw", {k: v for k, v in sd.items() if k not in ("quantizer.cluster_size", "quantizer.embed_avg")})
    log("Training completed!")
    codec.close()
    return {"history": history, "best_val_loss": best_val, "steps_per_epoch": steps_per_epoch, "world": world}


def main(argv=None):
    parser = argparse.ArgumentParser(description="EMA codebook training for the VQ-VAE leaf codec on MI355X.")
    sub = parser.add_subparsers(dest="command", required=True)
    p = sub.add_parser("train", help="Train the codebook (encoder/decoder frozen).")
    p.add_argument("--pack", required=True, help="VQWPACK1 weight pack (vqvdb_amd/weightpack.py)")
    p.add_argument("--mode", choices=("codebook", "full"), default="codebook",
                   help="codebook: EMA codebook only (encoder/decoder frozen); full: AdamW on every weight + EMA codebook (training.py)")
    p.add_argument("--lr", type=float, default=1e-4, help="full mode: AdamW learning rate (training.py:51)")
    p.add_argument("--data_dir", type=str, default=None, help="Directory with .npy leaf arrays [N,8,8,8]; synthetic leaves if omitted.")
    p.add_argument("--epochs", type=int, default=30)                      # training.py:50
    p.add_argument("--batch_size", type=int, default=2048, help="leaves per rank per step (training.py:49)")
    p.add_argument("--leaves_per_epoch", type=int, default=8_000_000, help="synthetic mode: training leaves per epoch (BASELINE configs[4])")
    p.add_argument("--commitment_cost", type=float, default=0.25)         # training.py:55
    p.add_argument("--decay", type=float, default=0.95)
    p.add_argument("--eps", type=float, default=1e-4)
    p.add_argument("--seed", type=int, default=0)
    p.add_argument("--log_every", type=int, default=100)
    p.add_argument("--model_path", type=str, default="models/quantizer.npz")
    p.add_argument("--resume", type=str, default=None, help="checkpoint (.npz written as --model_path) to continue from")
    p.add_argument("--backend", type=str, default="nccl", help="torch.distributed backend (nccl = RCCL)")
    p.add_argument("--single_gpu_rehearsal", action="store_true", help="tests: every rank on cuda:0 (use with --backend gloo)")
    p.set_defaults(func=train)
    args = parser.parse_args(argv)
    return args.func(args)


if __name__ == "__main__":
    main()
    sys.exit(0)
