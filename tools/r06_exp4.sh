#!/bin/bash
# round 6, experiment 4: the reduction stream on a hardware queue of its own (stream priority) against sharing the weight-gradient stream's queue
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; cd $R
{
python tools/train_ab.py
VQHIP_TRAIN_RED_PRIO=low python tools/train_ab.py
VQHIP_TRAIN_RED_PRIO=normal python tools/train_ab.py
VQHIP_TRAIN_BIAS=main VQHIP_TRAIN_EMA_AT=backward python tools/train_ab.py
VQHIP_TRAIN_SIDE_PRIO=low python tools/train_ab.py
VQHIP_TRAIN_SIDE_PRIO=high VQHIP_TRAIN_RED_PRIO=normal python tools/train_ab.py
VQHIP_TRAIN_SIDE_PRIO=low VQHIP_TRAIN_RED_PRIO=normal python tools/train_ab.py
python tools/train_ab.py
} > $O/r06_exp4_train_ab.txt 2>&1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $O/tl -o t -- python $R/tools/train_step_timeline.py --run 2048 > /dev/null 2>&1
python $R/tools/train_step_timeline.py $O/tl > $O/r06_train_step_timeline_2048_v3.txt 2>&1
rm -rf $O/tl
cd $R; grep -v amdgpu.ids $O/r06_exp4_train_ab.txt; grep "^queue\|^one" $O/r06_train_step_timeline_2048_v3.txt
