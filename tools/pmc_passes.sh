#!/bin/bash
# PMC passes over the step kernel (one small counter group per run); summaries under gpurun_out/pmc/.
set -e
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc; mkdir -p $OUT
i=0
for grp in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" \
           "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_INSTS_BRANCH" \
           "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_ANY" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_THREAD_CYCLES_VALU SQ_WAIT_ANY" \
           "SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_FLAT SQ_INSTS_FLAT"; do
  i=$((i+1))
  rm -rf /tmp/pmc_$i
  timeout -k 10 200 rocprofv3 --pmc $grp --output-format csv -d /tmp/pmc_$i -o p -- python3 $GRAFT_REPO_ROOT/bench.py --random-actions --steps 6 --warmup 2 --no-cpu-baseline --no-train-step > $OUT/run_$i.log 2>&1
  python3 $GRAFT_REPO_ROOT/tools/pmc_summary.py /tmp/pmc_$i vnl_step > $OUT/pass_$i.txt
  cat $OUT/pass_$i.txt
done
