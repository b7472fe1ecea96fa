#!/bin/bash
# One round of profiling evidence on the GPU box (run through gpurun from the repo root), in the order the bench line needs:
#   1. separate PMC passes (HBM fetch / write, SQ + GRBM, TCC) of a short bench -> profiles/hbm_traffic_per_launch.json of THIS build
#   2. the default bench (reads that table: roofline.traffic is current, traffic_stale = false) and the same bench under rocprofv3 --kernel-trace
# Outputs under gpurun_out/prof_sum/ (small text / json files for profiles/); the databases stay on the box.
set -x
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof
TAG=${PROFILE_TAG:-r05_v1}
SUM=$R/gpurun_out/prof_sum; rm -rf $OUT $SUM; mkdir -p $OUT $SUM
cd /tmp && export TMPDIR=/tmp
SHORT="--steps 4 --warmup 1 --no-cpu-baseline --no-train --no-host-path"
rocprofv3 --pmc FETCH_SIZE -d $OUT/pmc_fetch -o r02 -- python $R/bench.py $SHORT > /dev/null 2> $OUT/pmc_fetch.err
rocprofv3 --pmc WRITE_SIZE -d $OUT/pmc_write -o r02 -- python $R/bench.py $SHORT > /dev/null 2> $OUT/pmc_write.err
python $R/tools/make_traffic_json.py $OUT/pmc_fetch/r02_results.db $OUT/pmc_write/r02_results.db "profiles/${TAG}_bench_rocprofv3_summary.txt (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes, tools/profile_round.sh)" $SUM/hbm_traffic_per_launch.json
cp $SUM/hbm_traffic_per_launch.json $R/profiles/hbm_traffic_per_launch.json     # the bench below reads it from the tree it runs in
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY -d $OUT/pmc_sq -o r02 -- python $R/bench.py $SHORT > /dev/null 2> $OUT/pmc_sq.err
# L2 (TCC) requests / hits / misses per kernel: which part of a kernel's operand traffic the L2 absorbs (weights, re-read activations)
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum -d $OUT/pmc_tcc -o r02 -- python $R/bench.py $SHORT > /dev/null 2> $OUT/pmc_tcc.err
python $R/bench.py > $OUT/bench.json 2> $OUT/bench.err
rocprofv3 --kernel-trace --stats -d $OUT/kt -o r02 -- python $R/bench.py --steps 16 --warmup 2 --no-host-path > $OUT/bench_under_rocprof.json 2> $OUT/kt.err
find $OUT -name "*.db" | xargs ls -la
head -c 600 $OUT/bench.json
python $R/tools/rocpd_summary.py --kt $OUT/kt/r02_results.db --fetch $OUT/pmc_fetch/r02_results.db --write $OUT/pmc_write/r02_results.db > $SUM/${TAG}_bench_rocprofv3_summary.txt
python $R/tools/pmc_sq_summary.py $OUT/pmc_sq/r02_results.db > $SUM/${TAG}_pmc_mfma_util_clock.txt
python $R/tools/pmc_tcc_summary.py $OUT/pmc_tcc/r02_results.db $OUT/pmc_fetch/r02_results.db > $SUM/${TAG}_pmc_l2_hit_miss.txt 2> $SUM/${TAG}_pmc_l2.err || tail -3 $OUT/pmc_tcc.err
cp $OUT/bench.json $SUM/${TAG}_bench.json; cp $OUT/bench_under_rocprof.json $SUM/${TAG}_bench_under_rocprof.json
find $OUT -name "*.db" | xargs rm -f
ls -la $SUM
