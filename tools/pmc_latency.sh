#!/bin/bash
# Average latencies seen by the step kernel's waves (derived rocprofv3 counters) and the instruction-cache hit rate:
# one PMC pass per group.  usage (GPU box): bash tools/pmc_latency.sh > gpurun_out/pmc_latency.txt
cd /tmp && export TMPDIR=/tmp
for grp in "SmemLatency" "LdsLatency" "InstrFetchLatency" "VmemLatency" "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE" "SQ_IFETCH SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SMEM"; do
  d=/tmp/pmcl_$(echo $grp | tr ' ' '_' | cut -c1-40); rm -rf $d
  timeout -k 10 200 rocprofv3 --pmc $grp --output-format csv -d $d -o p -- python3 $GRAFT_REPO_ROOT/bench.py --random-actions --no-autoreset --steps 4 --warmup 1 --no-cpu-baseline --no-train-step > $d.log 2>&1 || { echo "group [$grp] failed"; tail -3 $d.log; continue; }
  echo "== $grp"
  python3 $GRAFT_REPO_ROOT/tools/pmc_summary.py $d vnl_step
done
