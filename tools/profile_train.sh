#!/bin/bash
# rocprofv3 kernel stats of one PPO training step (tools/train_bench.py); summary -> gpurun_out/prof_train/
set -e
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_train; mkdir -p $OUT
rm -rf /tmp/pt_stats
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pt_stats -o t -- python3 $GRAFT_REPO_ROOT/tools/train_bench.py --steps 2 ${TRAIN_ARGS} > $OUT/train_stats.log 2>&1
cp $(find /tmp/pt_stats -name "*kernel_stats.csv") $OUT/kernel_stats.csv
tail -1 $OUT/train_stats.log | cut -c1-300
python3 $GRAFT_REPO_ROOT/tools/kernel_timeline.py $(find /tmp/pt_stats -name "*kernel_trace.csv") 130 > $OUT/timeline_tail.txt
head -25 $OUT/kernel_stats.csv | cut -c1-60,150-260
