#!/bin/bash
# PMC passes over the PPO minibatch step's kernels (tools/ppo_update_bench.py); per-kernel averages -> gpurun_out/pmc_ppo/
set -e
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_ppo; mkdir -p $OUT
i=0
for grp in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
           "SQ_WAVE_CYCLES SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_ACTIVE_INST_LDS" \
           "SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_WAIT_INST_ANY"; do
  i=$((i+1))
  rm -rf /tmp/pmcp_$i
  timeout -k 10 200 rocprofv3 --pmc $grp --output-format csv -d /tmp/pmcp_$i -o p -- python3 $GRAFT_REPO_ROOT/tools/ppo_update_bench.py --iters 6 ${PPO_BENCH_ARGS:-} > $OUT/run_$i.log 2>&1 || true
  python3 $GRAFT_REPO_ROOT/tools/pmc_summary.py /tmp/pmcp_$i gemm_ > $OUT/pass_$i.txt || true
  cat $OUT/pass_$i.txt
done
