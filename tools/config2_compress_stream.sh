#!/bin/bash
# BASELINE config 2 at orchestrator level: 1M leaves (2 GB of fp32 voxels in leaf buffers) -> .vqvdb file.
#  (a) orchestrator-shaped compress (fresh pack buffer per batch, HipBackend::encode, serial framing)
#  (b) vqhip_compress_file (gather || GPU encode || framing + write)
set -e
cd "$(dirname "$0")/.."
python - <<'PY'
import numpy as np
from vqvdb_amd import synth, weightpack
weightpack.save("/tmp/model.vqw", synth.make_weights(0))
np.tile(synth.make_leaves(65536, seed=1234), (16, 1))[:1000000].tofile("/tmp/c2.f32")
PY
for i in 1 2; do ./vqvdb_amd/host/leaf_harness compress /tmp/model.vqw /tmp/c2.f32 /tmp/c2a.vqvdb 65536; done
for i in 1 2; do ./vqvdb_amd/host/leaf_harness compress_stream /tmp/model.vqw /tmp/c2.f32 /tmp/c2b.vqvdb 65536; done
cmp /tmp/c2a.vqvdb /tmp/c2b.vqvdb && echo "files identical"
