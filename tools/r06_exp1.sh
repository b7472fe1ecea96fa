#!/bin/bash
# round 6, experiment 1: raw first-conv loaders vs the row layout; host-call sub-chunking (VQHIP_HOST_SPLIT) on the leaf-pointer loops and the host path
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; cd $R
python -m pytest tests/test_gpu_parity.py -x -q -k "variants or small_batch_split or golden or all_layers or full_size or ragged or leaf_pointer" 2>&1 | tail -3 > $O/r06_exp1_tests.txt
{
python tools/bench_kernels.py
VQHIP_FIRST_SRC=packed python tools/bench_kernels.py
python tools/bench_kernels.py
VQHIP_FIRST_SRC=packed python tools/bench_kernels.py
VQHIP_FIRST=roll0 python tools/bench_kernels.py
} > $O/r06_exp1_kernels.txt 2>&1
python -c "
from vqvdb_amd import synth, weightpack
open('/tmp/m.vqw','wb').write(weightpack.dumps(synth.make_weights(0)))"
{
for sp in 1 2 3 4 6 8; do echo "== VQHIP_HOST_SPLIT=$sp"; VQHIP_HOST_SPLIT=$sp vqvdb_amd/host/leaf_harness loopbench_ptrs /tmp/m.vqw 1048576 /dev/shm/t.vqvdb 8192,32768,65536; done
for sp in 1 4; do echo "== kept loop VQHIP_HOST_SPLIT=$sp"; VQHIP_HOST_SPLIT=$sp vqvdb_amd/host/leaf_harness loopbench /tmp/m.vqw 1048576 /dev/shm/t.vqvdb 65536 0; done
} > $O/r06_exp1_host.txt 2>&1
cat $O/r06_exp1_tests.txt $O/r06_exp1_kernels.txt $O/r06_exp1_host.txt
