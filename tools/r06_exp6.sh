#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $O/tl -o t -- python $R/tools/train_step_timeline.py --run 2048 > /dev/null 2>&1
python $R/tools/train_step_timeline.py $O/tl > $O/r06_train_step_timeline_2048_v5.txt 2>&1
rm -rf $O/tl
grep "^queue\|^one" $O/r06_train_step_timeline_2048_v5.txt
