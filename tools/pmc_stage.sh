#!/bin/bash
# Instruction counts per stage: PMC over the step kernel with VNL_DBG_REPEAT=stage:4 (stage 0 = base).
set -e
# needs the diagnostic library: python vnl-brax-imitation_amd/csrc/build.py --knobs (before gpurun)
export VNL_LIB=$GRAFT_REPO_ROOT/vnl-brax-imitation_amd/csrc/libvnl_knobs.so
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_stage; mkdir -p $OUT
for st in ${STAGES:-0 1 2 3 4 5 6 7 8 9 10 11 12 13}; do
  if [ $st -ne 0 ]; then export VNL_DBG_REPEAT=$st:4; fi
  rm -rf /tmp/pmcs_$st
  timeout -k 10 200 rocprofv3 --pmc ${COUNTERS:-SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES} --output-format csv -d /tmp/pmcs_$st -o p -- python3 $GRAFT_REPO_ROOT/bench.py --random-actions --no-autoreset --steps 4 --warmup 1 --no-cpu-baseline > $OUT/run_$st.log 2>&1
  echo "stage $st" >> $OUT/all.txt
  python3 $GRAFT_REPO_ROOT/tools/pmc_summary.py /tmp/pmcs_$st vnl_step >> $OUT/all.txt
done
cat $OUT/all.txt
