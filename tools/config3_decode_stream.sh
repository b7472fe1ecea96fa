#!/bin/bash
# BASELINE config 3: 1xMI355X decode-only, .vqvdb -> leaves, 4M leaves streamed in 64k-leaf batches.
#  (a) orchestrator-shaped: IVQVAECodec::create -> HipBackend::decode per batch, fresh Tensor per batch, per-leaf
#      copies, everything serial (what the kept VQVAECodec::decompress does)
#  (b) vqhip_decompress_file: reader thread (file read + de-framing + leaf allocation) || GPU decode || scatter
#      into the leaf buffers; leaf store = hash map keyed by origin + 2 KiB per leaf (stand-in for tree.touchLeaf)
set -e
cd "$(dirname "$0")/.."
python - <<'PY'
from vqvdb_amd import synth, weightpack
weightpack.save("/tmp/model.vqw", synth.make_weights(0))
PY
N=${1:-4000000}
./vqvdb_amd/host/leaf_harness makefile /tmp/c3.vqvdb $N
ls -la /tmp/c3.vqvdb
for i in 1 2; do ./vqvdb_amd/host/leaf_harness decompress /tmp/model.vqw /tmp/c3.vqvdb /dev/null 65536; done
for i in 1 2; do ./vqvdb_amd/host/leaf_harness decompress_stream /tmp/model.vqw /tmp/c3.vqvdb /dev/null 65536; done
