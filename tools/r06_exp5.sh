#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; cd $R
python -m pytest tests/test_gpu_fulltrain.py tests/test_gpu_training.py -x -q 2>&1 | tail -4 > $O/r06_exp5_tests.txt
{
python tools/train_ab.py 2048 4096 8192 16384
VQHIP_TRAIN_RED_PRIO=own python tools/train_ab.py 2048 4096 8192 16384
VQHIP_TRAIN_RED_PRIO=normal python tools/train_ab.py 2048 4096 8192 16384
VQHIP_TRAIN_BIAS=main VQHIP_TRAIN_EMA_AT=backward python tools/train_ab.py 2048 4096 8192 16384
python tools/train_ab.py 2048 4096 8192 16384
} > $O/r06_exp5_train_ab.txt 2>&1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $O/tl -o t -- python $R/tools/train_step_timeline.py --run 2048 > /dev/null 2>&1
python $R/tools/train_step_timeline.py $O/tl > $O/r06_train_step_timeline_2048_v4.txt 2>&1
rm -rf $O/tl
cd $R; cat $O/r06_exp5_tests.txt; grep -v amdgpu.ids $O/r06_exp5_train_ab.txt; grep "^queue\|^one" $O/r06_train_step_timeline_2048_v4.txt
