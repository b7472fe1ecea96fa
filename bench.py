#!/usr/bin/env python3
"""Throughput bench of the VQ-VAE leaf hot path on MI355X (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W
    (N>1: either launched by `python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...`, one rank per GPU, or as the
     plain command above: with no WORLD_SIZE in the environment it starts its N ranks itself under torch.distributed.run on
     127.0.0.1 and a free port, passes the one JSON line of rank 0 through on its own stdout and returns the launcher's exit code)

A step = one pass of the hot path over one 65,536-leaf batch per GPU, inputs resident in HBM.
N = 1: BASELINE configs[1] ("1xMI355X, 1M synthetic leaves, fp32 encoder+quantizer" = 16 such batches, cycled for K steps).
N > 1: BASELINE configs[3] ("leaves sharded across GPUs, 64M leaves" = 8 Mi leaves = 128 batches per GPU): the timed region
       is max(K, 128) steps per rank so that every rank processes at least its configs[3] shard; `steps` in the JSON is the
       number of steps really timed.
`value` is encode+quantize leaves/s over all ranks; the decode leg is timed the same way and reported under "decode".
Leaves shard across ranks with no data-path collective (weak scaling: per-GPU work is fixed).  PyTorch is used only for
device memory, streams/events and the torch.distributed barrier.  Legs that go through host memory (end-to-end host-pointer
calls, BASELINE configs[2] .vqvdb streaming decode) run in a PyTorch-free child process: vqvdb_amd/hostbench.py.
"""
from __future__ import annotations

import argparse
import json
import os
import subprocess
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from vqvdb_amd import synth, weightpack  # noqa: E402
from vqvdb_amd.codec import HipCodec  # noqa: E402
from vqvdb_amd.sharding import bind_rank_to_cpus, gpu_numa_node, max_over_ranks, usable_cpus  # noqa: E402

BATCH = int(os.environ.get("VQ_BENCH_BATCH", "65536"))   # leaves per step and per GPU (the library's chunk); the environment override is for chunk-size experiments only
CONFIG3_LEAVES_PER_GPU = 8 * 1024 * 1024   # BASELINE configs[3]: 64 Mi leaves over 8 GPUs
ENC_FLOP = 30_589_952      # nominal dense FLOP / leaf
DEC_FLOP = 114_135_040
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
    return (max_over_ranks(dt, device) if dist else dt), dt


def profile_pass(codec, fn, steps, device):
    """Per-kernel averages from HIP events on the launch stream (library hook).  tflops_issued = FLOPs the kernel really
    executes on the matrix pipe / time; tflops_nominal_dense = dense FLOP count of the reference ops it replaces / time."""
    for s in range(4):          # back to steady state first (the pass may follow the small-kernel training legs: clocks, caches)
        fn(s)
    torch.cuda.synchronize(device)
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
        tf = (lambda f: round(f * leaves / (avg_ms * 1e-3) / 1e12, 2) if avg_ms > 0 else 0.0)
        out.append(dict(kernel=st["name"], avg_ms=round(avg_ms, 4), launches=st["launches"],
                        tflops_issued=tf(st["eff_flops_per_leaf"]), tflops_nominal_dense=tf(st["flops_per_leaf"]),
                        issued_flop_per_leaf=st["eff_flops_per_leaf"], nominal_flop_per_leaf=st["flops_per_leaf"]))
    return out


# FLOPs per leaf the full training step issues on the matrix pipe (2 x MACs; zero-padding taps skipped — tap fractions 0.7703 at 8^3 k3,
# 0.5787 at 4^3 k3, 0.6699 for k4 s2 — the decoder tail as ONE folded operator, forward and backward):
#   forward : first conv 170 k + 16->16 convs 2 x 2.726 M + down 1.405 M + 32->32 convs 2 x 1.024 M + projection 0.262 M + distances to
#             the 256 codes on the materialised latent 2.097 M + 64->64 convs 2 x 4.096 M + folded tail 1.573 M   = 21.20 M MAC
#             (the decoder stem runs through the (tap, code) table rebuilt every step: 56.6 M MAC per STEP, not per leaf — round 3;
#             as a real conv it was 8.192 M MAC per leaf)
#   dgrad   : every layer but the first conv (transposed convs on the forward kernels: the same MACs), folded tail through the
#             transposed operator 1.835 M                                                                    = 27.39 M MAC
#   wgrad   : every conv layer (dW = sum over leaves and positions of dY X^T: the forward's MACs), folded tail 1.835 M = 27.56 M MAC
TRAIN_ISSUED_FLOP = {"forward": 2 * 21.20e6, "dgrad": 2 * 27.39e6, "wgrad": 2 * 27.56e6}


def train_step_classes(codec, step_fn, leaves, device, steps=3):
    """Per kernel class of the full training step (HIP events of the library's profiler over `steps` steps): time, issued TFLOP/s and the
    fraction of the fp32-MFMA peak.  Classes by launch name: *_wgrad* weight gradients, *_dgrad* data gradients, the forward's
    MFMA kernels, everything else (statistics, elementwise GroupNorm / attention forward and backward, loss, optimizer, table rebuilds)."""
    fwd = ("enc_conv_first", "enc_res16_conv", "enc_down", "enc_res32_conv", "train_latent_assign", "ft_stem", "ft_res64_conv", "ft_tail", "ft_up_conv", "ft_final")
    torch.cuda.synchronize(device)
    codec.profile_enable(True)
    for _ in range(steps):
        step_fn()
    torch.cuda.synchronize(device)
    stats = codec.profile_read()
    codec.profile_enable(False)
    ms = {"forward": 0.0, "dgrad": 0.0, "wgrad": 0.0, "other": 0.0}
    for st in stats:
        n = st["name"]
        cls = "wgrad" if "_wgrad" in n else "dgrad" if "_dgrad" in n else "forward" if n.startswith(fwd) and n != "ft_tail_fold" else "other"
        ms[cls] += st["total_ms"] / steps
    out = {"summed_event_spans_ms_per_step": round(sum(ms.values()), 4)}   # (two streams: the spans overlap; not a wall time)
    for cls, t in ms.items():
        e = {"ms": round(t, 4)}
        if cls in TRAIN_ISSUED_FLOP and t > 0:
            tf = TRAIN_ISSUED_FLOP[cls] * leaves / (t * 1e-3) / 1e12
            e.update(tflops_issued=round(tf, 2), frac=round(tf / PEAK_TF, 4))
        out[cls] = e
    return out


def hbm_traffic(kernel):
    """HBM bytes per launch of `kernel` from the committed PMC passes (profiles/hbm_traffic_per_launch.json,
    tools/make_traffic_json.py): FETCH_SIZE (x2 gfx950 correction) + WRITE_SIZE.  None if not profiled."""
    try:
        t = json.load(open(os.path.join(ROOT, "profiles", "hbm_traffic_per_launch.json")))[kernel]
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        from make_traffic_json import kernel_build_id
        now = kernel_build_id()
        return {"bytes": t["fetch_bytes"] + t["write_bytes"], "fetch_bytes": t["fetch_bytes"], "write_bytes": t["write_bytes"],
                "collected_on_build": t.get("build"), "this_build": now, "stale": t.get("build") != now,
                "source": t["source"], "note": "separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this bench at 65536 leaves per launch; " + t["correction"]}
    except (OSError, KeyError, ValueError):
        return None


# bytes per leaf a layer reads + writes when every tensor crosses HBM exactly once (fp32, leaf-tile layout): input (+ residual input) + output
LAYER_IO_BYTES = {"pack_leaves": 2048 + 3072, "enc_conv_first_stats": 3072, "enc_conv_first_gn": 3072 + 32768, "enc_res16_conv1": 2 * 32768, "enc_res16_conv2": 3 * 32768,
                  "enc_down": 32768 + 8192, "enc_res32_conv1": 2 * 8192, "enc_res32_conv2": 3 * 8192, "enc_vq": 8192 + 64,
                  "dec_stem_gn": 64 + 16384, "dec_res64_conv1": 2 * 16384, "dec_res64_conv2": 3 * 16384, "dec_tail": 16384 + 2048, "dec_tail_slab": 16384 + 2048, "dec_tail_rows32": 16384 + 2048, "dec_tail_groups": 16384 + 2048}
ALGORITHMIC_BYTES = 2048 + 64      # per leaf, either direction: the leaf and its 64 indices


def path_traffic(kernels):
    """HBM bytes per 65536-leaf pass summed over the pass's kernels (committed PMC table), None when a kernel is missing from the table."""
    tot = 0
    for k in kernels:
        t = hbm_traffic(k["kernel"])
        if t is None:
            if k["kernel"] in LAYER_IO_BYTES:
                return None
            continue      # (statistics-combine launches: a few hundred KB)
        tot += t["bytes"]
    return tot


# FLOPs per leaf a kernel issues that the RESULT does not need (they are inside issued_flop_per_leaf): the first conv is run twice
# (statistics pass, then recompute + normalise + store: the first pass's MACs buy no output), and the folded decoder tail multiplies the
# structural zeros its 16-voxel tiles cannot skip (884 736 MAC/leaf are structurally non-zero; the corner tiles and the W axis are issued in full)
NOT_USEFUL_FLOP = {"enc_conv_first_stats": None, "dec_tail": lambda issued: issued - 2.0 * 884736, "dec_tail_slab": lambda issued: issued - 2.0 * 884736,
                   "dec_tail_rows32": lambda issued: issued - 2.0 * 884736, "dec_tail_groups": lambda issued: issued - 2.0 * 884736}


def useful_flop_per_leaf(kernels):
    tot = 0.0
    for k in kernels:
        f = k["issued_flop_per_leaf"]
        if k["kernel"] in NOT_USEFUL_FLOP:
            g = NOT_USEFUL_FLOP[k["kernel"]]
            f = 0.0 if g is None else f - g(f)
        tot += f
    return tot


def roofline_of(kernels, nominal_flop_per_leaf, leaves_per_s_per_gpu):
    """Roofline of the dominant kernel (longest average launch).  `achieved`/`frac` count the FLOPs the kernel ISSUES on the
    matrix pipe (zero-padding taps are skipped, folded operators counted at their folded cost) — a true utilisation, <= 1, that
    agrees with SQ_VALU_MFMA_BUSY x clock / 2.4 GHz from the PMC passes in profiles/.  The dense count of the reference ops the
    kernel replaces is reported beside it as *_nominal_dense; it can exceed the peak and is not a utilisation."""
    dom = max(kernels, key=lambda k: k["avg_ms"])
    issued_per_leaf = sum(k["issued_flop_per_leaf"] for k in kernels)
    tr = hbm_traffic(dom["kernel"])
    ptr = path_traffic(kernels)
    return {
        # flat copies of the traffic figures (a summariser that drops nested objects keeps these): the dominant kernel's HBM bytes per launch
        # against its layer's tensors crossing HBM once, and the whole pass against the algorithmic 2112 B per leaf
        "traffic_bytes": tr["bytes"] if tr else None,
        "traffic_over_layer_io": round(tr["bytes"] / (LAYER_IO_BYTES[dom["kernel"]] * BATCH), 3) if tr and dom["kernel"] in LAYER_IO_BYTES else None,
        "traffic_stale": tr["stale"] if tr else None,
        "whole_path_traffic_bytes": ptr,
        "whole_path_traffic_over_algorithmic": round(ptr / (ALGORITHMIC_BYTES * BATCH), 1) if ptr else None,
        "bound": "mfma", "kernel": dom["kernel"], "achieved": dom["tflops_issued"], "peak": PEAK_TF, "unit": "TFLOP/s",
        "frac": round(dom["tflops_issued"] / PEAK_TF, 4), "traffic": tr,
        "avg_launch_ms": dom["avg_ms"], "flop_per_launch_issued": dom["issued_flop_per_leaf"] * BATCH,
        "achieved_nominal_dense": dom["tflops_nominal_dense"], "ratio_nominal_dense_to_peak": round(dom["tflops_nominal_dense"] / PEAK_TF, 4),
        "whole_path_frac": round(leaves_per_s_per_gpu * issued_per_leaf / (PEAK_TF * 1e12), 4),
        "whole_path_issued_flop_per_leaf": issued_per_leaf,
        # ... counting only the FLOPs the result needs (no structural-zero MACs of the folded tail, the first conv once)
        "whole_path_frac_useful": round(leaves_per_s_per_gpu * useful_flop_per_leaf(kernels) / (PEAK_TF * 1e12), 4),
        "whole_path_useful_flop_per_leaf": useful_flop_per_leaf(kernels),
        "whole_path_ratio_nominal_dense_to_peak": round(leaves_per_s_per_gpu * nominal_flop_per_leaf / (PEAK_TF * 1e12), 4),
        "note": "achieved = FLOPs issued on the matrix pipe per launch / average launch time (HIP events on the launch stream); frac = achieved / peak; "
                "whole_path_frac = leaves/s x issued FLOP per leaf of every kernel of the pass / peak; *_nominal_dense = dense FLOP count of the reference ops "
                "(SURVEY App. A, padding taps and folded operators counted in full) over the same times — what the algebraic eliminations buy, not a utilisation",
    }


def cpu_baseline():
    """CPU oracle (the repo's C restatement of the reference path, kind 'port') on a bounded sample: the same uniform-random
    leaves as the GPU workload, at all usable cores and at half of them (the reference's default: TorchBackend.cpp:74-78
    sets at::set_num_threads(hardware_concurrency() / 2))."""
    from oracle.oracle import Oracle
    W = synth.make_weights(0)
    orc = Oracle(W, [t[0] for t in synth.TENSORS])
    threads = usable_cpus()
    n = max(2048, min(16384, 256 * threads))
    leaves = synth.make_leaves(n, seed=1234)

    def leg(fn, arg, nthreads, budget):
        t0 = time.perf_counter(); out = fn(arg, threads=nthreads); t1 = time.perf_counter() - t0
        reps = int(max(1, min(40, budget / max(t1, 1e-3))))
        t0 = time.perf_counter()
        for _ in range(reps):
            fn(arg, threads=nthreads)
        return out, reps, time.perf_counter() - t0

    idx, er, te = leg(orc.encode, leaves, threads, 8.0)
    _, dr, td = leg(orc.decode, idx, threads, 8.0)
    half = max(1, threads // 2)
    _, er2, te2 = leg(orc.encode, leaves, half, 5.0)
    _, dr2, td2 = leg(orc.decode, idx, half, 5.0)
    return {"value": round(n * er / te, 1), "unit": "leaves/s", "cores": threads, "kind": "port",
            "sample": f"{er} x {n} uniform-random leaves encode+quantize ({te:.1f} s) and {dr} x {n} decode ({td:.1f} s) at {threads} threads; "
                      f"{er2} x {n} encode ({te2:.1f} s) and {dr2} x {n} decode ({td2:.1f} s) at {half} threads; "
                      f"C oracle, OpenMP over 16-leaf tiles "
                      f"(usable CPUs: affinity {len(os.sched_getaffinity(0))}, cgroup quota applied; {os.cpu_count()} visible)",
            "decode_value": round(n * dr / td, 1),
            "cores_half": half, "value_half_cores": round(n * er2 / te2, 1), "decode_value_half_cores": round(n * dr2 / td2, 1),
            "half_cores_note": "the reference's libtorch CPU backend defaults to hardware_concurrency()/2 threads (src/backends/torch/TorchBackend.cpp:74-78)"}


def host_legs(args):
    """End-to-end host-memory legs in a PyTorch-free child process (vqvdb_amd/hostbench.py): host_path + BASELINE configs[2]."""
    cmd = [sys.executable, "-m", "vqvdb_amd.hostbench", "--file-leaves", str(args.file_leaves), "--reps", "3"]
    try:
        r = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=900)
        lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
        if r.returncode != 0 or not lines:
            return {"error": f"hostbench rc {r.returncode}: {(r.stderr or r.stdout)[-400:]}"}
        return json.loads(lines[-1])
    except Exception as e:  # noqa: BLE001 — never take the headline measurement down
        return {"error": f"{type(e).__name__}: {e}"}


class _RehearsalCodec:
    """CPU stand-in used ONLY by tests/test_sharding_gloo.py (VQ_BENCH_CPU_REHEARSAL=1) to exercise this file's N > 1 protocol —
    process group, barriers, max-over-ranks timing, per-rank gathers, the JSON shape — on a box without a GPU.  It computes
    nothing; a line produced this way says so in `data` and is not a measurement."""

    def set_chunk_leaves(self, n): pass
    def encode_device(self, *a): time.sleep(2e-4)
    def decode_device(self, *a): time.sleep(2e-4)
    def profile_enable(self, on): pass

    def profile_read(self):
        return [dict(name=n, launches=4, total_ms=4 * ms, flops_per_leaf=2 * f, eff_flops_per_leaf=f, leaves=4 * BATCH)
                for n, ms, f in (("rehearsal_kernel_a", 3.0, 5.0e6), ("rehearsal_kernel_b", 1.0, 1.0e6))]

    def close(self): pass


def self_launch(n):
    """`python bench.py --gpus N` (N > 1) started as ONE process: run the same command line as N ranks under torch.distributed.run
    (one process per GPU, rendezvous on 127.0.0.1 at a free port).  The ranks inherit this process's stdout — rank 0's JSON line
    is the only thing they write there (main() points every rank's fd 1 at stderr and keeps a duplicate for the line) — and
    the launcher's exit code is returned unchanged."""
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC: RCCL needs it on this host driver
    env.setdefault("OMP_NUM_THREADS", "1")              # what torchrun would set (and warn about) itself
    sys.stdout.flush()
    return subprocess.run(cmd, env=env).returncode


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=64)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-train", action="store_true", help="skip the training legs (BASELINE configs[4])")
    ap.add_argument("--no-host-path", action="store_true", help="skip the host-memory legs (host-pointer C ABI end to end; configs[2] .vqvdb streaming decode)")
    ap.add_argument("--file-leaves", type=int, default=4 * 1024 * 1024, help="leaves in the configs[2] .vqvdb file")
    args = ap.parse_args()
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ and os.environ.get("VQ_BENCH_FORCE_DIST") != "1":
        raise SystemExit(self_launch(args.gpus))

    # Only the final JSON line may reach stdout.  Native libraries (RCCL prints its version banner with
    # NCCL_DEBUG=VERSION, which this image exports) write to fd 1 directly, so fd 1 is pointed at stderr
    # for the whole run and the JSON goes to a saved duplicate of the original stdout.
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    local_world = int(os.environ.get("LOCAL_WORLD_SIZE", str(world)))
    # Rehearsal of the N > 1 code path on a box with ONE GPU (tests/tools only): every rank uses cuda:0 and the process
    # group runs on gloo (RCCL refuses two ranks on one device).  The numbers of such a run are meaningless.
    rehearsal = os.environ.get("VQ_BENCH_SINGLE_GPU_REHEARSAL") == "1"
    # ... and on a box with NO GPU (CPU tests of the N > 1 protocol only): a stand-in codec that computes nothing, gloo, CPU tensors
    cpu_rehearsal = os.environ.get("VQ_BENCH_CPU_REHEARSAL") == "1"
    if cpu_rehearsal:
        rehearsal = True
        args.no_cpu_baseline = args.no_train = args.no_host_path = True
        torch.cuda.synchronize = lambda *a, **k: None
    dev_index = 0 if rehearsal else local
    dist = world > 1 or os.environ.get("VQ_BENCH_FORCE_DIST") == "1"   # the latter exercises the RCCL path with one rank
    if world != args.gpus and not os.environ.get("VQ_BENCH_FORCE_DIST"):
        raise SystemExit(f"--gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks")
    device = torch.device("cpu") if cpu_rehearsal else torch.device("cuda", dev_index)
    if not cpu_rehearsal:
        torch.cuda.set_device(device)
    # one rank = one GPU = its own slice of the host's cores (NUMA node of the GPU when the topology is readable): the host-side
    # copy threads of N ranks must not pile onto the same cores
    affinity = None
    if world > 1 and not rehearsal:
        try:
            # NUMA node of EVERY local rank's GPU (local rank r drives device r), read from /sys by each rank itself: ranks that share a
            # node then split it in rank order even when the GPU-to-node mapping is interleaved
            props = [torch.cuda.get_device_properties(i) for i in range(min(local_world, torch.cuda.device_count()))]
            nodes = [gpu_numa_node(getattr(p, "pci_bus_id", None), getattr(p, "pci_domain_id", 0)) for p in props]
            affinity = bind_rank_to_cpus(local, local_world, pci_bus_id=getattr(props[local], "pci_bus_id", None),
                                         pci_domain_id=getattr(props[local], "pci_domain_id", 0),
                                         gpu_numa_nodes=nodes if len(nodes) == local_world and None not in nodes else None)
        except Exception as e:  # noqa: BLE001 — binding is an optimisation, never a reason to fail
            affinity = {"error": f"{type(e).__name__}: {e}"}
    elif world > 1 and cpu_rehearsal:      # the CPU protocol rehearsal binds too (no PCI topology: the even split), so the plan itself is exercised
        affinity = bind_rank_to_cpus(local, local_world)
    if dist:
        if rehearsal:
            torch.distributed.init_process_group("gloo")
        else:
            torch.distributed.init_process_group("nccl", device_id=device)
    # what the collective library really saw: an all-reduce of ones over the group, and each rank's device
    dev_name = "cpu (rehearsal)" if cpu_rehearsal else torch.cuda.get_device_name(device)
    dev_props = None if cpu_rehearsal else torch.cuda.get_device_properties(device)
    backend, ranks_seen, devices_seen = None, 1, [dev_name]
    if dist:
        backend = torch.distributed.get_backend()
        one = torch.ones(1, device=device if backend == "nccl" else "cpu")
        torch.distributed.all_reduce(one)
        ranks_seen = int(one.item())
        objs = [None] * torch.distributed.get_world_size()
        torch.distributed.all_gather_object(objs, {"rank": rank, "local_rank": local, "device": dev_index, "device_by_local_rank": local, "name": dev_name,
                                                   "cpu_set": sorted(os.sched_getaffinity(0)) if affinity and affinity.get("bound") else None,
                                                   "pci_bus_id": getattr(dev_props, "pci_bus_id", None),
                                                   "cpus": (affinity or {}).get("cpus_bound")})
        devices_seen = objs
        assert ranks_seen == world == torch.distributed.get_world_size(), (ranks_seen, world)

    W = synth.make_weights(0)
    codec = _RehearsalCodec() if cpu_rehearsal else HipCodec(weightpack.dumps(W), device_id=dev_index)
    codec.set_chunk_leaves(BATCH)

    # N > 1: every rank covers at least its configs[3] shard (8 Mi leaves = 128 batches)
    steps = args.steps if world == 1 else max(args.steps, CONFIG3_LEAVES_PER_GPU // BATCH)

    # synthetic leaves, uniform [0,1) (training data is [0,1]-normalised), resident in HBM; each
    # rank gets its own shard (different seed) -> no data-path collective.
    nb = 1 if cpu_rehearsal else min(steps, 16)
    rows = 64 if cpu_rehearsal else BATCH                     # the stand-in codec touches no memory
    gen = torch.Generator(device=device)
    gen.manual_seed(1234 + rank)
    leaves = [torch.rand(rows, 512, device=device, dtype=torch.float32, generator=gen) for _ in range(nb)]
    idx = [torch.empty(rows, 64, device=device, dtype=torch.uint8) for _ in range(nb)]
    rec = torch.empty(rows, 512, device=device, dtype=torch.float32)
    stream = 0 if cpu_rehearsal else torch.cuda.current_stream(device).cuda_stream

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
    t_enc, t_enc_local = timed(enc, steps, dist, device)
    t_dec, t_dec_local = timed(dec, steps, dist, device)
    enc_lps = world * steps * BATCH / t_enc
    dec_lps = world * steps * BATCH / t_dec
    per_rank = None
    if dist:   # every rank's own rate (its own clock around its own steps): min / max show stragglers
        objs = [None] * world
        torch.distributed.all_gather_object(objs, (steps * BATCH / t_enc_local, steps * BATCH / t_dec_local))
        per_rank = {"encode_leaves_per_s": [round(o[0], 1) for o in objs], "decode_leaves_per_s": [round(o[1], 1) for o in objs],
                    "encode_min_max": [round(min(o[0] for o in objs), 1), round(max(o[0] for o in objs), 1)],
                    "decode_min_max": [round(min(o[1] for o in objs), 1), round(max(o[1] for o in objs), 1)],
                    "note": "leaves/s of each rank over its own wall time for the timed steps (barriers included); value uses the slowest rank's time"}

    # Small-batch latency (position-split kernels): what the SOP's per-batch calls see (default 64, max 1024 encode / 8192
    # decode leaves, SOP_VQVDB_Encoder.cpp:33-38).  Device-resident, rank 0 only.
    small = None
    if rank == 0 and not cpu_rehearsal:
        small = {"note": "ms per call on the device at SOP-sized batches (position-split path); not the headline value"}
        for nsm in (64, 1024, 8192):
            for _ in range(3):
                codec.encode_device(leaves[0].data_ptr(), nsm, idx[0].data_ptr(), stream)
                codec.decode_device(idx[0].data_ptr(), nsm, rec.data_ptr(), stream)
            t_e, _ = timed(lambda s: codec.encode_device(leaves[0].data_ptr(), nsm, idx[0].data_ptr(), stream), 10, False, device)
            t_d, _ = timed(lambda s: codec.decode_device(idx[0].data_ptr(), nsm, rec.data_ptr(), stream), 10, False, device)
            small[f"leaves_{nsm}"] = {"encode_ms": round(t_e / 10 * 1e3, 4), "decode_ms": round(t_d / 10 * 1e3, 4),
                                      "encode_leaves_per_s": round(nsm * 10 / t_e, 1), "decode_leaves_per_s": round(nsm * 10 / t_d, 1)}
        enc(0)   # idx[0] back to the full batch's indices for the parity spot check below

    coll = (lambda nfl: f"all_reduce(SUM) of {nfl} fp32 over {world} rank(s), {'RCCL' if backend == 'nccl' else backend}") if dist else (lambda nfl: "none (1 rank)")

    # Codebook (EMA) training steps, the quantizer part of BASELINE configs[4]: per-rank batches of 2048 leaves (the
    # reference's BATCH_SIZE, python/training.py:49) and of 65536 leaves; statistics all-reduced over RCCL when N > 1.
    train = None
    if not args.no_train:
        try:
            from vqvdb_amd.codebook_training import CodebookTrainer
            tcodec = HipCodec(weightpack.dumps(W), device_id=dev_index)   # its own handle: training rewrites the codebook
            tcodec.set_chunk_leaves(BATCH)
            trainer = CodebookTrainer(tcodec, device=str(device))
            train = {"note": "VectorQuantizerEMA training-mode step on encoder outputs (encoder forward + latent + assign + "
                             "statistics + all-reduce + EMA update); encoder/decoder weights frozen; not the headline value",
                     "collective": coll(33281)}
            ksteps = max(2, min(args.steps, 8))
            for per_rank_b in (2048, BATCH):
                x = leaves[0][:per_rank_b]
                for _ in range(2):
                    trainer.step(x, want_metrics=False)
                t_tr, _ = timed(lambda s: trainer.step(x, want_metrics=False), ksteps, dist, device)
                last = trainer.step(x)
                train[f"per_rank_batch_{per_rank_b}"] = {"leaves_per_s": round(world * ksteps * per_rank_b / t_tr, 1),
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
            fcodec = HipCodec(weightpack.dumps(W), device_id=dev_index)
            ftr = FullTrainer(fcodec, device=str(device))
            full = {"note": "one optimizer step = training-mode forward (decoder stem through the per-step (tap, code) table, decoder tail folded) + backward of every layer + all-reduce + AdamW + EMA "
                            "codebook update + rebuild of the weight-derived tables, fp32; not the headline value",
                    "collective": coll("995905 + 33284"),
                    "flop_per_leaf_nominal": 3 * (ENC_FLOP + DEC_FLOP), "flop_per_leaf_issued": TRAIN_ISSUED_FLOP,
                    "frac_note": "whole_step_frac and by_class[*].frac count the FLOPs the step ISSUES on the matrix pipe (padding taps skipped, folded tail at its "
                                 "folded cost; forward + data gradients + weight gradients, TRAIN_ISSUED_FLOP in bench.py) over the measured time / 157.3 TFLOP/s: "
                                 "true utilisations, <= 1.  ratio_nominal_dense_* = 3 x the dense forward count of the reference ops over the same time: what the "
                                 "algebraic eliminations buy, not a utilisation (it exceeds 1 at 8192 leaves per rank)",
                    "streams": 1 if os.environ.get("VQHIP_TRAIN_STREAMS") == "1" else 2,
                    "streams_note": "two streams: the data-gradient chain on the caller's stream, ending with the step's deferred reductions (bias sums, GroupNorm-affine "
                                    "and attention-weight reductions as two multi-job launches, round 6); weight gradients, the codebook statistics (started beside the "
                                    "decoder's forward) and the tail fold on the second. by_class times are event spans on either stream, they overlap, and "
                                    "summed_event_spans_ms_per_step exceeds ms_per_step; whole_step_frac (from the wall time) is the utilisation of the step"}
            ksteps = max(2, min(args.steps, 6))
            for per_rank_b in (2048, 8192):
                x = leaves[0][:per_rank_b]
                for _ in range(2):
                    ftr.step(x, want_metrics=False)
                t_ft, _ = timed(lambda s: ftr.step(x, want_metrics=False), ksteps, dist, device)
                last = ftr.step(x)
                lps = world * ksteps * per_rank_b / t_ft
                entry = {"leaves_per_s": round(lps, 1), "ms_per_step": round(t_ft / ksteps * 1e3, 4), "steps": ksteps,
                         "loss": round(last["loss"], 6), "perplexity": round(last["perplexity"], 3),
                         "whole_step_frac": round(lps / world * sum(TRAIN_ISSUED_FLOP.values()) / (PEAK_TF * 1e12), 4),
                         "ratio_nominal_dense_to_fp32_mfma_peak": round(lps / world * 3 * (ENC_FLOP + DEC_FLOP) / (PEAK_TF * 1e12), 4)}
                # (every rank runs the profiled steps: a step contains the gradient all-reduce)
                by_class = train_step_classes(fcodec, lambda: ftr.step(x, want_metrics=False), per_rank_b, device)
                if rank == 0:
                    entry["by_class"] = by_class
                full[f"per_rank_batch_{per_rank_b}"] = entry
            fcodec.close()
        except Exception as e:  # noqa: BLE001 — never take the headline measurement down
            full = {"error": f"{type(e).__name__}: {e}"}

    if rank == 0:
        ek = profile_pass(codec, enc, min(steps, 16), device)
        dk = profile_pass(codec, dec, min(steps, 16), device)
        # spot parity: first 64 leaves of batch 0 vs the CPU oracle
        parity = None
        cpu = None
        if not args.no_cpu_baseline:
            cpu = cpu_baseline()
            from oracle.oracle import Oracle
            orc = Oracle(W, [t[0] for t in synth.TENSORS])
            h = leaves[0][:64].cpu().numpy()
            enc(0)
            torch.cuda.synchronize(device)
            gi = idx[0][:64].cpu().numpy()
            oi = orc.encode(h, threads=usable_cpus())
            parity = f"{int((gi == oi).all(axis=1).sum())}/64 sampled leaves index-exact vs CPU oracle"
        host = None
        if world == 1 and not args.no_host_path:
            # free this process's big buffers first: the child holds its own workspace and an 8 GB leaf pool
            torch.cuda.synchronize(device)
            host = host_legs(args)
        elif world > 1:
            host = {"skipped": "host-memory legs run at N = 1 only (one process per GPU already saturates its PCIe link)"}
        enc_roof, dec_roof = roofline_of(ek, ENC_FLOP, enc_lps / world), roofline_of(dk, DEC_FLOP, dec_lps / world)
        dtail = next((k for k in dk if k["kernel"].startswith("dec_tail")), None)
        # the decode half of the metric inside `roofline` too (the object a summariser keeps whole): BASELINE's metric is
        # "encode+quantize AND decode"; `value` is the encode leg, these are the decode leg of the same run
        enc_roof.update({
            "decode_value": round(dec_lps, 1), "decode_unit": "leaves/s", "decode_ms_per_step": round(t_dec / steps * 1e3, 4),
            "decode_kernel": dec_roof["kernel"], "decode_frac": dec_roof["frac"], "decode_achieved": dec_roof["achieved"],
            "decode_avg_launch_ms": dec_roof["avg_launch_ms"],
            "decode_whole_path_frac": dec_roof["whole_path_frac"], "decode_whole_path_frac_useful": dec_roof["whole_path_frac_useful"],
            "dec_tail_ms": dtail["avg_ms"] if dtail else None,
            "decode_traffic_bytes": dec_roof["whole_path_traffic_bytes"], "decode_traffic_stale": dec_roof["traffic_stale"],
            "decode_workload": "decode of the same 65536-leaf index batches, resident in HBM, timed like `value` (K steps, barrier + synchronize on both sides)",
        })
        out = {
            "metric": "8^3 leaves/s encode+quantize (decode reported under 'decode')",
            "value": round(enc_lps, 1), "unit": "leaves/s", "n_gpus": world, "steps": steps, "warmup": args.warmup,
            "ms_per_step": round(t_enc / steps * 1e3, 4), "timed_region_s": round(t_enc, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic" if not cpu_rehearsal else "REHEARSAL on CPU with a stand-in codec: protocol test only, not a measurement",
            "config": {"workload": (f"BASELINE configs[1]: 1xMI355X, 1M synthetic leaves (16 x 65536-leaf batches, cycled for {steps} steps), fp32 encoder+quantizer, K=256 D=128; decode leg: the same batches' indices -> voxels, same K steps (roofline.decode_*)" if world == 1 else
                                    f"BASELINE configs[3]: {world}xMI355X encode, leaves sharded across GPUs, {steps * BATCH} leaves per GPU "
                                    f"(>= the 8 Mi-leaf shard of the 64M-leaf job; 65536-leaf batches), fp32 encoder+quantizer, K=256 D=128"),
                       "leaves_per_step_per_gpu": BATCH, "leaves_per_gpu": steps * BATCH, "steps_requested": args.steps,
                       "sharding": f"leaves sharded over {world} rank(s), no data-path collective"},
            # flat copies of the figures a summariser that drops nested objects would lose (the nested objects below stay authoritative)
            "decode_value": round(dec_lps, 1), "decode_ms_per_step": round(t_dec / steps * 1e3, 4),
            "roofline_frac": enc_roof["frac"], "decode_roofline_frac": dec_roof["frac"],
            "whole_path_frac": enc_roof["whole_path_frac"], "decode_whole_path_frac": dec_roof["whole_path_frac"],
            "whole_path_frac_useful": enc_roof["whole_path_frac_useful"], "decode_whole_path_frac_useful": dec_roof["whole_path_frac_useful"],
            "dec_tail_ms": dtail["avg_ms"] if dtail else None, "dec_tail_issued_flop_per_leaf": dtail["issued_flop_per_leaf"] if dtail else None,
            "traffic_stale": enc_roof["traffic_stale"], "decode_traffic_stale": dec_roof["traffic_stale"],
            "n_devices_seen": len(devices_seen),
            "per_rank_encode_min": per_rank["encode_min_max"][0] if per_rank else None, "per_rank_encode_max": per_rank["encode_min_max"][1] if per_rank else None,
            "per_rank_decode_min": per_rank["decode_min_max"][0] if per_rank else None, "per_rank_decode_max": per_rank["decode_min_max"][1] if per_rank else None,
            "collective_backend": ("nccl (RCCL)" if backend == "nccl" else backend) if dist else None,
            "ranks_seen": ranks_seen,
            "devices_seen": devices_seen,
            "per_rank": per_rank,
            "cpu_affinity": affinity,
            "roofline": enc_roof,
            "cpu_baseline": cpu,
            "decode": {
                "value": round(dec_lps, 1), "unit": "leaves/s", "ms_per_step": round(t_dec / steps * 1e3, 4), "timed_region_s": round(t_dec, 4),
                "workload": "decode of 65536-leaf index batches resident in HBM (kernel path of BASELINE configs[2]; the file-level run is under 'config3')",
                "roofline": dec_roof,
            },
            "kernels": {"encode": ek, "decode": dk},
            "flop_per_leaf": {"encode_nominal_dense": ENC_FLOP, "decode_nominal_dense": DEC_FLOP,
                              "encode_issued": sum(k["issued_flop_per_leaf"] for k in ek), "decode_issued": sum(k["issued_flop_per_leaf"] for k in dk)},
            "parity_sample": parity,
            "host_path": (host or {}).get("host_path") if host and "host_path" in host else host,
            "config3": (host or {}).get("config3"),
            "orchestrator_loop": (host or {}).get("orchestrator_loop"),
            "host_process": (host or {}).get("process"),
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
