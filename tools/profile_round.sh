#!/bin/bash
# One round of profiling evidence on the GPU box (run through gpurun from the repo root):
#   kernel trace + stats of the default bench, then separate PMC passes (HBM fetch / write, SQ+GRBM) of a short bench.
# Outputs under gpurun_out/prof/; summarise with tools/rocpd_summary.py, tools/pmc_sq_summary.py, tools/make_traffic_json.py.
set -x
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
python $R/bench.py --steps 16 --warmup 2 > $OUT/bench.json 2> $OUT/bench.err
rocprofv3 --kernel-trace --stats -d $OUT/kt -o r01 -- python $R/bench.py --steps 16 --warmup 2 > $OUT/bench_under_rocprof.json 2> $OUT/kt.err
SHORT="--steps 4 --warmup 1 --no-cpu-baseline --no-train"
rocprofv3 --pmc FETCH_SIZE -d $OUT/pmc_fetch -o r01 -- python $R/bench.py $SHORT > /dev/null 2> $OUT/pmc_fetch.err
rocprofv3 --pmc WRITE_SIZE -d $OUT/pmc_write -o r01 -- python $R/bench.py $SHORT > /dev/null 2> $OUT/pmc_write.err
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY -d $OUT/pmc_sq -o r01 -- python $R/bench.py $SHORT > /dev/null 2> $OUT/pmc_sq.err
find $OUT -name "*.db" | xargs ls -la
cat $OUT/bench.json | head -c 600
