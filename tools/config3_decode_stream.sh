#!/bin/bash
# BASELINE config 3: 1xMI355X decode-only, .vqvdb -> leaves, 4M leaves streamed in 64k-leaf batches,
# through the C++ adapter (IVQVAECodec::create -> HipBackend::decode) exactly as the orchestrator calls it.
set -e
cd "$(dirname "$0")/.."
python - <<'PY'
from vqvdb_amd import synth, weightpack
weightpack.save("/tmp/model.vqw", synth.make_weights(0))
PY
N=${1:-4000000}
./vqvdb_amd/host/leaf_harness makefile /tmp/c3.vqvdb $N
ls -la /tmp/c3.vqvdb
./vqvdb_amd/host/leaf_harness decompress /tmp/model.vqw /tmp/c3.vqvdb /dev/null 65536
./vqvdb_amd/host/leaf_harness decompress /tmp/model.vqw /tmp/c3.vqvdb /dev/null 65536
