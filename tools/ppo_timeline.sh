#!/bin/bash
# Kernel timeline of ONE graph replay of the PPO minibatch step (tools/ppo_update_bench.py), fused and layered forward.
set -e
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/ppo_timeline; mkdir -p $OUT
for mode in ${PPO_MODES:-0 2}; do
  rm -rf /tmp/ppt_$mode
  extra="--fwd-mode $mode ${PPO_BENCH_ARGS:-}"
  timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d /tmp/ppt_$mode -o t -- python3 $GRAFT_REPO_ROOT/tools/ppo_update_bench.py --iters 3 $extra > $OUT/run_$mode.log 2>&1
  python3 $GRAFT_REPO_ROOT/tools/kernel_timeline.py $(find /tmp/ppt_$mode -name "*kernel_trace.csv") 48 > $OUT/timeline_$mode.txt
done
