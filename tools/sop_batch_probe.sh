#!/bin/bash
# Orchestrator-shaped compress / decompress (IVQVAECodec::create -> HipBackend::encode/decode per batch, serial, like
# VQVAECodec.cpp:78-208) at the SOP's batch sizes (default 64, max 1024 encode / 8192 decode), 200k leaves.
set -e
cd "$(dirname "$0")/.."
python - <<'PY'
import numpy as np
from vqvdb_amd import synth, weightpack
weightpack.save("/tmp/model.vqw", synth.make_weights(0))
np.tile(synth.make_leaves(8192, seed=1234), (25, 1))[:200000].tofile("/tmp/sop.f32")
PY
for b in 64 1024 8192 65536; do
  ./vqvdb_amd/host/leaf_harness compress /tmp/model.vqw /tmp/sop.f32 /tmp/sop.vqvdb $b
  ./vqvdb_amd/host/leaf_harness decompress /tmp/model.vqw /tmp/sop.vqvdb /dev/null $b
done
