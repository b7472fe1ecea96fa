"""End-to-end legs of the bench that go through HOST memory, in a process that never imports PyTorch.

    python -m vqvdb_amd.hostbench [--leaves N] [--file-leaves M] [--batch 65536] [--reps 3]

A process that imports PyTorch first binds libvqvdb_hip.so to the HIP runtime bundled with the wheel (ROCm 7.0) instead of
the system one (ROCm 7.2) and loses 10-15 % on the host-pointer path, and a Houdini plug-in is not
such a process — so bench.py runs these legs here, as a child, and embeds the JSON line this module prints:

  host_path : vqhip_encode / vqhip_decode, pageable numpy buffers in AND out (PCIe both ways included) — SURVEY §8(d)
              "end-to-end" figure next to the kernels-only headline value.
  config3   : BASELINE configs[2] — a single-grid `.vqvdb` of M (default 4 Mi) leaves decoded by vqhip_decompress_file in
              64k-leaf batches: file read + de-framing || GPU decode || scatter into per-leaf buffers handed out by the caller's
              allocator (here: consecutive 2 KiB slots of one pre-touched pool — the cheapest possible `touchLeaf`; the C++
              harness leg with a hash-map leaf store is reported beside it when the harness binary is present).
              Reference path replaced: src/orchestrator/VQVAECodec.cpp:137-208, src/Utils/VQVDB_Reader.cpp:240-335.
  orchestrator_loop : what a drop-in user of the KEPT orchestrator gets — the reference's two serial loops
              (VQVAECodec.cpp:78-134 compress, :137-208 decompress: IVQVAECodec::create -> per batch {fresh buffer -> TensorView ->
              encode/decode -> owning Tensor -> framing / per-leaf copies}) through the C++ adapter, 1 Mi leaves at the SOP's batch
              sizes (64 default, 1024 = encoder maximum, 8192 = decoder maximum; SOP_VQVDB_Encoder.cpp:33-38) and at 65536,
              with the per-call time split into its phases (`leaf_harness loopbench`).
"""
from __future__ import annotations

import argparse
import json
import os
import re
import subprocess
import sys
import tempfile
import time

import numpy as np

from . import synth, vqvdbfile, weightpack
from .codec import HipCodec

BATCH = 65536


def origins_of(n: int, start: int = 0) -> np.ndarray:
    """Distinct leaf origins (multiples of 8) in file order."""
    i = np.arange(start, start + n, dtype=np.int64)
    return np.stack([8 * (i % 1024), 8 * ((i // 1024) % 1024), 8 * (i // 1048576)], axis=1).astype(np.int32)


def write_index_file(path: str, indices_fn, n: int, chunk: int = 1 << 20, name: str = "density") -> None:
    """Single-grid .vqvdb v3 of n leaves written chunk by chunk (never more than `chunk` records in memory)."""
    with open(path, "wb") as f:
        head = vqvdbfile.dumps([vqvdbfile.Grid(name, np.zeros((0, 3), np.int32), np.zeros((0, 64), np.uint8))])
        f.write(head[:-4] + np.uint32(n).tobytes())          # header + grid metadata with totalBlocks = n
        for s in range(0, n, chunk):
            m = min(chunk, n - s)
            rec = np.empty(m, dtype=vqvdbfile.RECORD)
            rec["origin"] = origins_of(m, s)
            rec["indices"] = indices_fn(s, m)
            f.write(rec.tobytes())


LOOP_RE = re.compile(r"loopbench(?:-mt|-ptrs)? (compress|decompress)\s+batch (\d+): (\d+) leaves in ([\d.]+) ms = ([\d.]+) M leaves/s \| first call ([\d.]+) ms \| "
                     r"per call: [\w+]+ ([\d.]+) ms, (?:encode|decode) ([\d.]+) ms, [\w+ ]+ ([\d.]+) ms")


def orchestrator_loop(harness: str, W: dict, tmpdir: str, n: int, batches=(64, 1024, 8192, 65536)) -> dict:
    """`leaf_harness loopbench`: the reference orchestrator's serial compress / decompress loops through the adapter."""
    with tempfile.NamedTemporaryFile(suffix=".vqw", delete=False) as f:
        f.write(weightpack.dumps(W))
        pk = f.name
    tmp = os.path.join(tmpdir, f"vqhip_loop_{os.getpid()}.vqvdb")
    res = {"workload": f"{n} leaves per leg; reference call sequence (VQVAECodec.cpp:78-134,137-208) through IVQVAECodec::create + HipBackend, serial, "
                       "pageable host buffers, a fresh pack buffer / zero-filled Tensor per batch; batch sizes: SOP default 64, encoder max 1024, decoder max 8192 "
                       "(SOP_VQVDB_Encoder.cpp:33-38), and the backend's chunk 65536",
           "leaves": n, "batches": {}}
    try:
        def one(extra, into, mode="loopbench"):
            r = subprocess.run([harness, mode, pk, str(n), tmp, ",".join(str(b) for b in batches)] + extra, capture_output=True, text=True, timeout=900)
            if r.returncode != 0:
                return {"error": (r.stderr or r.stdout)[-400:]}
            for m in LOOP_RE.finditer(r.stdout):
                leg, b = m.group(1), m.group(2)
                names = ("pack_ms", "encode_ms", "frame_write_ms") if leg == "compress" else ("read_deframe_ms", "decode_ms", "leaf_copy_ms")
                into.setdefault(b, {})[leg] = {
                    "leaves_per_s": round(float(m.group(5)) * 1e6, 1), "wall_s": round(float(m.group(4)) / 1e3, 4), "first_call_ms": float(m.group(6)),
                    "per_call": dict(zip(names, (float(m.group(7)), float(m.group(8)), float(m.group(9)))))}
            return None if into else {"error": "no loopbench lines parsed: " + r.stdout[-300:]}
        err = one([], res["batches"])
        if err:
            return err
        # ... and with the pack loop and the leaf copies on half the cores, as tbb::parallel_for runs them in the reference (VQVAECodec.cpp:50,182)
        threads = max(1, (os.cpu_count() or 2) // 2)
        res["threaded"] = {"threads": threads, "note": "same loops, pack / leaf-copy ranges of >= 1024 leaves split over a fixed pool of std::threads "
                           "(hardware_concurrency() / 2); batches of up to 1024 leaves run on the caller either way", "batches": {}}
        err = one([str(threads)], res["threaded"]["batches"])
        if err:
            res["threaded"] = err
        # ... and the loops re-written on the leaf-pointer entry points: what the extension buys over the kept loop
        res["leaf_pointers"] = {"note": "the same two loops on vqhip_encode_leaves / vqhip_decode_leaves: the caller hands over per-leaf pointers (one 2 KiB heap block per "
                                        "leaf, shuffled, like OpenVDB leaf buffers); no pack buffer, no owning Tensor, no per-leaf copies on the caller's side — the library "
                                        "gathers / scatters with its own threads through pinned staging, overlapped with the GPU; per_call.pack_ms = building the pointer array, "
                                        "leaf_copy_ms = 0 by construction", "batches": {}}
        err = one([], res["leaf_pointers"]["batches"], "loopbench_ptrs")
        if err:
            res["leaf_pointers"] = err
    finally:
        os.unlink(pk)
        if os.path.exists(tmp):
            os.unlink(tmp)
    return res


def run(args) -> dict:
    W = synth.make_weights(0)
    codec = HipCodec(weightpack.dumps(W))
    codec.set_chunk_leaves(args.batch)
    codec.reserve(args.batch)
    out = {"process": "no PyTorch imported: libvqvdb_hip.so runs on the system ROCm runtime, like a plug-in would"}

    # ---- host-pointer entry points, pageable in and out ----
    n = args.leaves
    base = synth.make_leaves(min(n, BATCH), seed=1234)
    leaves = np.tile(base, (-(-n // len(base)), 1))[:n]
    idx = np.zeros((n, 64), np.uint8)
    rec = np.zeros((n, 512), np.float32)         # reused caller buffers (first-touch page faults outside the timed calls)
    codec.encode(leaves[:args.batch])
    te, td = [], []
    for _ in range(args.reps):
        t0 = time.perf_counter(); codec.encode(leaves, out=idx); te.append(time.perf_counter() - t0)
    codec.decode(idx[:args.batch])
    for _ in range(args.reps):
        t0 = time.perf_counter(); codec.decode(idx, out=rec); td.append(time.perf_counter() - t0)
    out["host_path"] = {
        "note": "vqhip_encode / vqhip_decode: pageable host memory in and out (H2D + kernels + D2H overlapped inside the library), "
                "PCIe included; median of the timed calls; never the headline value",
        "leaves": n, "reps": args.reps,
        "encode_leaves_per_s": round(n / sorted(te)[len(te) // 2], 1), "decode_leaves_per_s": round(n / sorted(td)[len(td) // 2], 1),
        "encode_s": [round(t, 4) for t in te], "decode_s": [round(t, 4) for t in td]}
    del leaves, rec

    # ---- BASELINE configs[2]: .vqvdb -> leaves, file_leaves leaves in 64k-leaf batches ----
    m = args.file_leaves
    tmpdir = "/dev/shm" if os.path.isdir("/dev/shm") and os.access("/dev/shm", os.W_OK) else tempfile.gettempdir()
    path = os.path.join(tmpdir, f"vqhip_config3_{os.getpid()}.vqvdb")
    try:
        rng = np.random.default_rng(7)
        write_index_file(path, lambda s, k: rng.integers(0, 256, size=(k, 64), dtype=np.uint8), m)
        pool = np.empty((m, 512), np.float32)
        pool.fill(0.0)                              # touch every page: the caller's tree owns its leaves before the call
        runs = []
        for _ in range(args.reps):
            grids, st = codec.decompress_file(path, batch_leaves=args.batch, out=pool)
            assert st["leaves"] == m and len(grids) == 1
            runs.append(st)
        runs.sort(key=lambda s: s["wall_s"])
        st = runs[len(runs) // 2]
        c3 = {"workload": f"BASELINE configs[2]: single-grid .vqvdb ({os.path.getsize(path) / 1e6:.0f} MB, in {tmpdir}) -> {m} leaves, {args.batch}-leaf batches, "
                          "vqhip_decompress_file (reader thread || GPU decode || scatter into caller leaf buffers)",
              "leaves": m, "batch_leaves": args.batch, "reps": args.reps,
              "leaves_per_s": round(m / st["wall_s"], 1), "wall_s": round(st["wall_s"], 4),
              "leaf_store": "consecutive 2 KiB slots of one pre-touched pool handed out per batch by a Python allocator callback",
              "stream_stats": {k: (round(v, 4) if isinstance(v, float) else v) for k, v in st.items()},
              "all_wall_s": [round(r["wall_s"], 4) for r in runs]}
        # sanity: the decoded voxels of the last timed run are sigmoid outputs
        samp = pool[:: max(1, m // 4096)]
        c3["sample_check"] = bool(np.isfinite(samp).all() and samp.min() > 0.0 and samp.max() < 1.0)
        del pool
        # the same file through the C++ harness (hash-map leaf store standing in for tree.touchLeaf), if it was built
        harness = os.path.join(os.path.dirname(os.path.abspath(__file__)), "host", "leaf_harness")
        if os.path.exists(harness) and not args.no_harness:
            with tempfile.NamedTemporaryFile(suffix=".vqw", delete=False) as f:
                f.write(weightpack.dumps(W))
                pk = f.name
            try:
                codec.close()
                codec = None
                r = subprocess.run([harness, "decompress_stream", pk, path, "/dev/null", str(args.batch)], capture_output=True, text=True, timeout=600)
                mm = re.search(r"([\d.]+) ms wall = ([\d.]+) M leaves/s .*leaf alloc ([\d.]+) ms.*waited for reader ([\d.]+) ms", r.stdout)
                if r.returncode == 0 and mm:
                    c3["cpp_harness_hashmap_leaf_store"] = {"leaves_per_s": round(float(mm.group(2)) * 1e6, 1), "wall_s": round(float(mm.group(1)) / 1e3, 4),
                                                            "leaf_alloc_s": round(float(mm.group(3)) / 1e3, 4), "pipeline_waited_for_reader_s": round(float(mm.group(4)) / 1e3, 4),
                                                            "note": "includes nothing but the call; first call of a fresh process (workspace reserved before)"}
                else:
                    c3["cpp_harness_hashmap_leaf_store"] = {"error": (r.stderr or r.stdout)[-300:]}
            finally:
                os.unlink(pk)
        out["config3"] = c3
        if os.path.exists(harness) and not args.no_harness and args.loop_leaves > 0:
            out["orchestrator_loop"] = orchestrator_loop(harness, W, tmpdir, args.loop_leaves)
    finally:
        if os.path.exists(path):
            os.unlink(path)
    if codec is not None:
        codec.close()
    return out


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--leaves", type=int, default=8 * BATCH)
    ap.add_argument("--file-leaves", type=int, default=4 * 1024 * 1024)
    ap.add_argument("--batch", type=int, default=BATCH)
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--no-harness", action="store_true")
    ap.add_argument("--loop-leaves", type=int, default=1 << 20, help="leaves per orchestrator_loop leg (0 = skip)")
    args = ap.parse_args(argv)
    assert "torch" not in sys.modules, "hostbench must run without PyTorch in the process"
    res = run(args)
    sys.stdout.write(json.dumps(res) + "\n")


if __name__ == "__main__":
    main()
