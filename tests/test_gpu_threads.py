"""Two codec handles driven from two caller threads (SURVEY.md §8(b) threading row).

The reference keeps one codec per SOP node cache (src/SOP/SOP_VQVDB_Encoder.hpp:43-50) and Houdini may cook an encoder node and a
decoder node at once: two independent handles, two host threads, no shared state but the device.  include/vqvdb_hip.h promises
"distinct handles may be used from distinct threads concurrently"; these tests hold it to that, bit for bit against the serial run."""
import os
import subprocess
import threading

import numpy as np
import pytest

from vqvdb_amd import synth, weightpack
from vqvdb_amd.codec import HipCodec

pytestmark = pytest.mark.gpu

SIZES = (64, 1, 333, 1024, 4096, 20000, 97, 8192, 2048, 12345)


def _bits(a):
    return np.where(a == 0, 0.0, a).astype(np.float32).view(np.uint32)


@pytest.fixture(scope="module")
def truth(weights):
    pack = weightpack.dumps(weights)
    leaves = synth.make_leaves(max(SIZES), seed=606)
    c = HipCodec(pack)
    idx = {n: c.encode(leaves[:n]) for n in SIZES}
    rec = {n: c.decode(idx[n]) for n in SIZES}
    c.close()
    return pack, leaves, idx, rec


def _run(threads):
    errs = []

    def wrap(fn):
        def go():
            try:
                fn()
            except BaseException as e:  # noqa: BLE001 — reported by the test below
                errs.append(repr(e))
        return go
    ts = [threading.Thread(target=wrap(f)) for f in threads]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    return errs


def test_encode_on_one_handle_while_decode_on_another(truth):
    """Thread A loops encode on handle 1, thread B loops decode on handle 2: 200 iterations each, batch sizes 1 ... 20 000, both launch
    paths (the position-split kernels of small passes and, every other iteration, the full-chunk kernels via set_small_batch_tiles(0))."""
    pack, leaves, idx, rec = truth
    h1, h2 = HipCodec(pack), HipCodec(pack)
    bad = []

    def enc():
        for i in range(200):
            n = SIZES[i % len(SIZES)]
            h1.set_small_batch_tiles(0 if i & 1 else -1)
            if not np.array_equal(h1.encode(leaves[:n]), idx[n]):
                bad.append(("encode", i, n))

    def dec():
        for i in range(200):
            n = SIZES[(3 * i + 1) % len(SIZES)]
            h2.set_small_batch_tiles(0 if (i >> 1) & 1 else -1)
            if not np.array_equal(_bits(h2.decode(idx[n])), _bits(rec[n])):
                bad.append(("decode", i, n))

    errs = _run([enc, dec])
    h1.close()
    h2.close()
    assert not errs and not bad, (errs, bad[:5])


def test_create_destroy_churn_beside_a_busy_handle(truth):
    """50 x create / use / destroy on thread B while thread A keeps encoding on its own handle."""
    pack, leaves, idx, rec = truth
    h1 = HipCodec(pack)
    bad, stop, count = [], threading.Event(), [0]

    def enc():
        i = 0
        while not stop.is_set():
            n = SIZES[i % len(SIZES)]
            if not np.array_equal(h1.encode(leaves[:n]), idx[n]):
                bad.append(("encode", i, n))
            i += 1
        count[0] = i

    def churn():
        try:
            for i in range(50):
                h = HipCodec(pack)
                n = SIZES[i % 4]
                if not np.array_equal(_bits(h.decode(idx[n])), _bits(rec[n])):
                    bad.append(("decode", i, n))
                h.close()
        finally:
            stop.set()

    errs = _run([enc, churn])
    h1.close()
    assert not errs and not bad and count[0] > 0, (errs, bad[:5], count)


def test_two_hipbackend_objects_two_threads_in_cpp(truth, tmp_path):
    """The same through the C++ adapter: two HipBackend objects from IVQVAECodec::create, one caller thread each, then create / destroy
    churn beside a busy backend (leaf_harness threads)."""
    from vqvdb_amd.build import build_harness
    pack = truth[0]
    (tmp_path / "m.vqw").write_bytes(pack)
    r = subprocess.run([build_harness(), "threads", str(tmp_path / "m.vqw"), "200"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and ": 0 mismatches" in r.stdout, r.stdout + r.stderr
