#!/usr/bin/env python3
"""Throughput bench of the VQ-VAE leaf hot path on MI355X (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W
    (N>1: python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...)

A step = one pass of the hot path over one 65,536-leaf batch per GPU, inputs resident in HBM
(config 2 of BASELINE.json: "1xMI355X, 1M synthetic leaves, fp32 encoder+quantizer", i.e. 16
steps of 64k leaves).  `value` is encode+quantize leaves/s over all ranks; the decode leg is
timed the same way and reported under "decode".  Leaves shard across ranks with no data-path
collective (weak scaling: per-GPU work is fixed).  PyTorch is used only for device memory,
streams/events and the torch.distributed barrier.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from vqvdb_amd import synth, weightpack  # noqa: E402
from vqvdb_amd.codec import HipCodec  # noqa: E402
from vqvdb_amd.sharding import max_over_ranks  # noqa: E402

BATCH = 65536
ENC_FLOP = 30_589_952      # nominal dense FLOP / leaf
DEC_FLOP = 114_135_040
ENC_FLOP_EFF = 22_869_760  # zero-padding taps excluded (informational)
DEC_FLOP_EFF = 66_221_568
PEAK_TF = 157.3            # fp32 MFMA dense peak per MI355X (MI355X_MICROARCH.md)


def timed(fn, steps, dist, device):
    if dist:
        torch.distributed.barrier()
    torch.cuda.synchronize(device)
    t0 = time.perf_counter()
    for s in range(steps):
        fn(s)
    torch.cuda.synchronize(device)
    if dist:
        torch.distributed.barrier()
    dt = time.perf_counter() - t0
    return max_over_ranks(dt, device) if dist else dt


def profile_pass(codec, fn, steps, device, flop_key):
    """Per-kernel averages from HIP events on the launch stream (library hook)."""
    codec.profile_enable(True)
    for s in range(steps):
        fn(s)
    torch.cuda.synchronize(device)
    stats = codec.profile_read()
    codec.profile_enable(False)
    out = []
    for st in stats:
        avg_ms = st["total_ms"] / max(st["launches"], 1)
        leaves = st["leaves"] / max(st["launches"], 1)
        out.append(dict(kernel=st["name"], avg_ms=round(avg_ms, 4), launches=st["launches"],
                        tflops=round(st[flop_key] * leaves / (avg_ms * 1e-3) / 1e12, 2) if avg_ms > 0 else 0.0,
                        tflops_effective=round(st["eff_flops_per_leaf"] * leaves / (avg_ms * 1e-3) / 1e12, 2) if avg_ms > 0 else 0.0))
    return out


def hbm_traffic(kernel):
    """HBM bytes per launch of `kernel` from the committed PMC passes (profiles/hbm_traffic_per_launch.json,
    tools/make_traffic_json.py): FETCH_SIZE (x2 gfx950 correction) + WRITE_SIZE.  None if not profiled."""
    try:
        t = json.load(open(os.path.join(ROOT, "profiles", "hbm_traffic_per_launch.json")))[kernel]
        return {"bytes": t["fetch_bytes"] + t["write_bytes"], "fetch_bytes": t["fetch_bytes"], "write_bytes": t["write_bytes"],
                "source": t["source"], "note": "separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this bench at 65536 leaves per launch; " + t["correction"]}
    except (OSError, KeyError, ValueError):
        return None


def roofline_of(kernels, total_flop_per_leaf, leaves_per_s_per_gpu):
    dom = max(kernels, key=lambda k: k["avg_ms"])
    return {
        "bound": "mfma", "kernel": dom["kernel"], "achieved": dom["tflops"], "peak": PEAK_TF, "unit": "TFLOP/s",
        "frac": round(dom["tflops"] / PEAK_TF, 4), "traffic": hbm_traffic(dom["kernel"]),
        "achieved_effective": dom["tflops_effective"], "avg_launch_ms": dom["avg_ms"],
        "whole_path_frac": round(leaves_per_s_per_gpu * total_flop_per_leaf / (PEAK_TF * 1e12), 4),
        "note": "achieved = nominal dense FLOP per leaf of the reference ops this kernel replaces (SURVEY App. A, padding taps counted) "
                "x 65536 leaves / avg launch time; the kernels skip zero-padding taps (and fold/look up linear operators), so nominal can "
                "exceed the MFMA peak; achieved_effective = FLOPs really issued on the matrix pipe / time (true utilisation); "
                "whole_path_frac = leaves/s x nominal FLOP per leaf of the whole path / peak",
    }


def usable_cpus() -> int:
    """CPUs this process may really use: min(affinity mask, cgroup v2 cpu.max quota)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = max(1, min(n, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return n


def cpu_baseline():
    """CPU oracle (the repo's C restatement of the reference path, kind 'port') on a bounded sample:
    the same uniform-random leaves as the GPU workload, repeated until ~10 s of CPU work per leg."""
    from oracle.oracle import Oracle
    W = synth.make_weights(0)
    orc = Oracle(W, [t[0] for t in synth.TENSORS])
    threads = usable_cpus()
    n = max(4096, min(16384, 256 * threads))
    leaves = synth.make_leaves(n, seed=1234)

    def leg(fn, arg, budget=10.0):
        t0 = time.perf_counter(); out = fn(arg, threads=threads); t1 = time.perf_counter() - t0
        reps = int(max(1, min(40, budget / max(t1, 1e-3))))
        t0 = time.perf_counter()
        for _ in range(reps):
            fn(arg, threads=threads)
        return out, reps, time.perf_counter() - t0

    idx, er, te = leg(orc.encode, leaves)
    _, dr, td = leg(orc.decode, idx)
    return {"value": round(n * er / te, 1), "unit": "leaves/s", "cores": threads, "kind": "port",
            "sample": f"{er} x {n} uniform-random leaves encode+quantize ({te:.1f} s) and {dr} x {n} decode ({td:.1f} s); "
                      f"C oracle, OpenMP over 16-leaf tiles, {threads} threads "
                      f"(= usable CPUs: affinity {len(os.sched_getaffinity(0))}, cgroup quota applied; {os.cpu_count()} visible)",
            "decode_value": round(n * dr / td, 1)}, (leaves, idx)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=16)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-train", action="store_true", help="skip the codebook-training leg (BASELINE configs[4], quantizer part)")
    ap.add_argument("--host-path", action="store_true", help="also time the host-pointer C ABI (pageable memory in/out, PCIe included)")
    args = ap.parse_args()

    # Only the final JSON line may reach stdout.  Native libraries (RCCL prints its version banner with
    # NCCL_DEBUG=VERSION, which this image exports) write to fd 1 directly, so fd 1 is pointed at stderr
    # for the whole run and the JSON goes to a saved duplicate of the original stdout.
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    # Rehearsal of the N > 1 code path on a box with ONE GPU (tests/tools only): every rank uses cuda:0 and the process
    # group runs on gloo (RCCL refuses two ranks on one device).  The numbers of such a run are meaningless.
    rehearsal = os.environ.get("VQ_BENCH_SINGLE_GPU_REHEARSAL") == "1"
    if rehearsal:
        local = 0
    dist = world > 1 or os.environ.get("VQ_BENCH_FORCE_DIST") == "1"   # the latter exercises the RCCL path with one rank
    if dist:
        if rehearsal:
            torch.distributed.init_process_group("gloo")
        else:
            torch.distributed.init_process_group("nccl", device_id=torch.device("cuda", local))
    elif args.gpus != 1:
        raise SystemExit("--gpus N>1 must be launched with torch.distributed.run (one process per GPU)")
    device = torch.device("cuda", local)
    torch.cuda.set_device(device)

    W = synth.make_weights(0)
    codec = HipCodec(weightpack.dumps(W), device_id=local)
    codec.set_chunk_leaves(BATCH)

    # synthetic leaves, uniform [0,1) (training data is [0,1]-normalised), resident in HBM; each
    # rank gets its own shard (different seed) -> no data-path collective.
    nb = min(args.steps, 16)
    gen = torch.Generator(device=device)
    gen.manual_seed(1234 + rank)
    leaves = [torch.rand(BATCH, 512, device=device, dtype=torch.float32, generator=gen) for _ in range(nb)]
    idx = [torch.empty(BATCH, 64, device=device, dtype=torch.uint8) for _ in range(nb)]
    rec = torch.empty(BATCH, 512, device=device, dtype=torch.float32)
    stream = torch.cuda.current_stream(device).cuda_stream

    def enc(s):
        codec.encode_device(leaves[s % nb].data_ptr(), BATCH, idx[s % nb].data_ptr(), stream)

    def dec(s):
        codec.decode_device(idx[s % nb].data_ptr(), BATCH, rec.data_ptr(), stream)

    for s in range(max(args.warmup, 1)):
        enc(s)
    for s in range(nb):            # make sure every index buffer is populated for the decode leg
        enc(s)
    for s in range(max(args.warmup, 1)):
        dec(s)
    t_enc = timed(enc, args.steps, dist, device)
    t_dec = timed(dec, args.steps, dist, device)
    enc_lps = world * args.steps * BATCH / t_enc
    dec_lps = world * args.steps * BATCH / t_dec

    # Small-batch latency (position-split kernels): what the SOP's per-batch calls see (default 64, max 1024 encode / 8192
    # decode leaves, SOP_VQVDB_Encoder.cpp:33-38).  Device-resident, rank 0 only.
    small = None
    if rank == 0:
        small = {"note": "ms per call on the device at SOP-sized batches (position-split path); not the headline value"}
        for nsm in (64, 1024, 8192):
            for _ in range(3):
                codec.encode_device(leaves[0].data_ptr(), nsm, idx[0].data_ptr(), stream)
                codec.decode_device(idx[0].data_ptr(), nsm, rec.data_ptr(), stream)
            t_e = timed(lambda s: codec.encode_device(leaves[0].data_ptr(), nsm, idx[0].data_ptr(), stream), 10, False, device)
            t_d = timed(lambda s: codec.decode_device(idx[0].data_ptr(), nsm, rec.data_ptr(), stream), 10, False, device)
            small[f"leaves_{nsm}"] = {"encode_ms": round(t_e / 10 * 1e3, 4), "decode_ms": round(t_d / 10 * 1e3, 4),
                                      "encode_leaves_per_s": round(nsm * 10 / t_e, 1), "decode_leaves_per_s": round(nsm * 10 / t_d, 1)}
        enc(0)   # idx[0] back to the full batch's indices for the parity spot check below

    # Codebook (EMA) training steps, the quantizer part of BASELINE configs[4]: per-rank batches of 2048 leaves (the
    # reference's BATCH_SIZE, python/training.py:49) and of 65536 leaves; statistics all-reduced over RCCL when N > 1.
    train = None
    if not args.no_train:
        try:
            from vqvdb_amd.codebook_training import CodebookTrainer
            tcodec = HipCodec(weightpack.dumps(W), device_id=local)   # its own handle: training rewrites the codebook
            tcodec.set_chunk_leaves(BATCH)
            trainer = CodebookTrainer(tcodec, device=str(device))
            train = {"note": "VectorQuantizerEMA training-mode step on encoder outputs (encoder forward + latent + assign + "
                             "statistics + all-reduce + EMA update); encoder/decoder weights frozen; not the headline value",
                     "collective": (f"all_reduce(SUM) of 33281 fp32 over {world} rank(s), "
                                    f"{'RCCL' if torch.distributed.get_backend() == 'nccl' else torch.distributed.get_backend()}") if dist else "none (1 rank)"}
            ksteps = max(2, min(args.steps, 8))
            for per_rank in (2048, BATCH):
                x = leaves[0][:per_rank]
                for _ in range(2):
                    trainer.step(x, want_metrics=False)
                t_tr = timed(lambda s: trainer.step(x, want_metrics=False), ksteps, dist, device)
                last = trainer.step(x)
                train[f"per_rank_batch_{per_rank}"] = {"leaves_per_s": round(world * ksteps * per_rank / t_tr, 1),
                                                        "ms_per_step": round(t_tr / ksteps * 1e3, 4), "steps": ksteps,
                                                        "vq_loss": round(last["vq_loss"], 6), "perplexity": round(last["perplexity"], 3)}
            tcodec.close()
        except Exception as e:  # noqa: BLE001 — the training leg must never take the headline measurement down
            train = {"error": f"{type(e).__name__}: {e}"}

    # Full training step (BASELINE configs[4]: fp32 VQ-VAE, AdamW on encoder/decoder, EMA codebook; global batch N x 2048):
    # forward + backward + all-reduce of gradients (3.98 MB) and statistics (133 KB) + optimizer step.
    full = None
    if not args.no_train:
        try:
            from vqvdb_amd.full_training import FullTrainer
            fcodec = HipCodec(weightpack.dumps(W), device_id=local)
            ftr = FullTrainer(fcodec, device=str(device))
            full = {"note": "one optimizer step = training-mode forward (unfolded decoder) + backward of every layer + all-reduce + AdamW + EMA "
                            "codebook update + rebuild of the weight-derived tables, fp32; not the headline value",
                    "collective": (f"all_reduce(SUM) of 995905 + 33284 fp32 over {world} rank(s), "
                                   f"{'RCCL' if torch.distributed.get_backend() == 'nccl' else torch.distributed.get_backend()}") if dist else "none (1 rank)",
                    "flop_per_leaf_nominal": 3 * (ENC_FLOP + DEC_FLOP)}
            ksteps = max(2, min(args.steps, 6))
            for per_rank in (2048, 8192):
                x = leaves[0][:per_rank]
                for _ in range(2):
                    ftr.step(x, want_metrics=False)
                t_ft = timed(lambda s: ftr.step(x, want_metrics=False), ksteps, dist, device)
                last = ftr.step(x)
                lps = world * ksteps * per_rank / t_ft
                full[f"per_rank_batch_{per_rank}"] = {"leaves_per_s": round(lps, 1), "ms_per_step": round(t_ft / ksteps * 1e3, 4), "steps": ksteps,
                                                       "loss": round(last["loss"], 6), "perplexity": round(last["perplexity"], 3),
                                                       "frac_of_fp32_mfma_peak_nominal": round(lps / world * 3 * (ENC_FLOP + DEC_FLOP) / (PEAK_TF * 1e12), 4)}
            fcodec.close()
        except Exception as e:  # noqa: BLE001 — never take the headline measurement down
            full = {"error": f"{type(e).__name__}: {e}"}

    if rank == 0:
        ek = profile_pass(codec, enc, min(args.steps, 4), device, "flops_per_leaf")
        dk = profile_pass(codec, dec, min(args.steps, 4), device, "flops_per_leaf")
        # spot parity: first 64 leaves of batch 0 vs the CPU oracle
        parity = None
        cpu = None
        if not args.no_cpu_baseline:
            cpu, _ = cpu_baseline()
            from oracle.oracle import Oracle
            orc = Oracle(W, [t[0] for t in synth.TENSORS])
            h = leaves[0][:64].cpu().numpy()
            enc(0)
            torch.cuda.synchronize(device)
            gi = idx[0][:64].cpu().numpy()
            oi = orc.encode(h, threads=usable_cpus())
            parity = f"{int((gi == oi).all(axis=1).sum())}/64 sampled leaves index-exact vs CPU oracle"
        host = None
        if args.host_path:
            nh = 8 * BATCH
            hl = np.tile(leaves[0].cpu().numpy(), (8, 1))
            codec.encode(hl[:BATCH])
            t0 = time.perf_counter(); hi = codec.encode(hl); te = time.perf_counter() - t0
            codec.decode(hi[:BATCH])
            t0 = time.perf_counter(); codec.decode(hi); td = time.perf_counter() - t0
            host = {"note": "host-pointer entry points (vqhip_encode/vqhip_decode), pageable host memory in and out, PCIe included; "
                            "never the headline value", "leaves": nh, "encode_leaves_per_s": round(nh / te, 1), "decode_leaves_per_s": round(nh / td, 1)}
        out = {
            "metric": "8^3 leaves/s encode+quantize (decode reported under 'decode')",
            "value": round(enc_lps, 1), "unit": "leaves/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(t_enc / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": ("BASELINE configs[1]: 1xMI355X, 1M synthetic leaves in 65536-leaf batches, fp32 encoder+quantizer, K=256 D=128" if world == 1 else
                                    f"BASELINE configs[3] shape: {world}xMI355X encode, leaves sharded across GPUs (65536-leaf batches per GPU per step), "
                                    "fp32 encoder+quantizer, K=256 D=128"),
                       "leaves_per_step_per_gpu": BATCH, "sharding": f"leaves sharded over {world} rank(s), no collective"},
            "roofline": roofline_of(ek, ENC_FLOP, enc_lps / world),
            "cpu_baseline": cpu,
            "decode": {
                "value": round(dec_lps, 1), "unit": "leaves/s", "ms_per_step": round(t_dec / args.steps * 1e3, 4),
                "workload": "BASELINE configs[2] kernel path: decode of 65536-leaf index batches resident in HBM",
                "roofline": roofline_of(dk, DEC_FLOP, dec_lps / world),
            },
            "kernels": {"encode": ek, "decode": dk},
            "flop_per_leaf": {"encode_nominal": ENC_FLOP, "encode_effective": ENC_FLOP_EFF, "decode_nominal": DEC_FLOP, "decode_effective": DEC_FLOP_EFF},
            "parity_sample": parity,
            "host_path": host,
            "codebook_training": train,
            "full_training": full,
            "small_batch": small,
        }
        os.write(json_fd, (json.dumps(out) + "\n").encode())
    if dist:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
