#!/bin/bash
# round 6, experiment 3: reduction stream + early codebook statistics in the training step; fixed raw first-conv loaders
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; cd $R
python -m pytest tests/test_gpu_fulltrain.py tests/test_gpu_training.py -x -q 2>&1 | tail -8 > $O/r06_exp3_tests.txt
python -m pytest tests/test_gpu_parity.py -x -q -k "variants_agree or all_layers" 2>&1 | tail -5 >> $O/r06_exp3_tests.txt
{
python tools/train_ab.py
VQHIP_TRAIN_BIAS=main VQHIP_TRAIN_EMA_AT=backward python tools/train_ab.py
VQHIP_TRAIN_BIAS=main python tools/train_ab.py
VQHIP_TRAIN_EMA_AT=backward python tools/train_ab.py
python tools/train_ab.py
VQHIP_TRAIN_SIDE_PRIO=low python tools/train_ab.py
VQHIP_TRAIN_SIDE_PRIO=high python tools/train_ab.py
} > $O/r06_exp3_train_ab.txt 2>&1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $O/tl -o t -- python $R/tools/train_step_timeline.py --run 2048 > /dev/null 2>&1
python $R/tools/train_step_timeline.py $O/tl > $O/r06_train_step_timeline_2048_v2.txt 2>&1
rm -rf $O/tl
cd $R; cat $O/r06_exp3_tests.txt $O/r06_exp3_train_ab.txt; head -3 $O/r06_train_step_timeline_2048_v2.txt
