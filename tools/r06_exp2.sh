#!/bin/bash
# round 6, experiment 2: why the raw first-conv loaders failed the intermediate comparison; the training step's timeline today; finer host splits
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; cd $R
VQHIP_FIRST_SRC=raw python -m pytest tests/test_gpu_parity.py -x -q -k "all_layers" 2>&1 | tail -40 > $O/r06_exp2_rawtest.txt
python -m pytest tests/test_gpu_parity.py -x -q -k "variants or all_layers or leaf_pointer or host" 2>&1 | tail -5 > $O/r06_exp2_tests.txt
python -c "
from vqvdb_amd import synth, weightpack
open('/tmp/m.vqw','wb').write(weightpack.dumps(synth.make_weights(0)))"
{
for sp in 8 16,4096 32,2048; do echo "== VQHIP_HOST_SPLIT=$sp"; VQHIP_HOST_SPLIT=$sp vqvdb_amd/host/leaf_harness loopbench_ptrs /tmp/m.vqw 1048576 /dev/shm/t.vqvdb 8192,16384,32768,65536; done
} > $O/r06_exp2_host.txt 2>&1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $O/tl -o t -- python $R/tools/train_step_timeline.py --run 2048 > /dev/null 2>&1
python $R/tools/train_step_timeline.py $O/tl > $O/r06_train_step_timeline_2048.txt 2>&1
rm -rf $O/tl
FT_N=2048 bash $R/tools/scratch/ft_trace.sh
cp $O/ft_kernels_2048.txt $O/r06_train_kernels_one_stream_2048.txt
cd $R; cat $O/r06_exp2_rawtest.txt | tail -30; cat $O/r06_exp2_tests.txt; head -5 $O/r06_train_step_timeline_2048.txt
