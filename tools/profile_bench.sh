#!/bin/bash
# rocprofv3 kernel stats + HBM traffic (separate PMC passes) of the default bench; summaries -> gpurun_out/prof_bench/
set -e
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_bench; mkdir -p $OUT
rm -rf /tmp/pb_stats /tmp/pb_rd /tmp/pb_wr
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pb_stats -o b -- python3 $GRAFT_REPO_ROOT/bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-train-step --no-free-running > $OUT/bench_stats.log 2>&1
cp $(find /tmp/pb_stats -name "*kernel_stats.csv") $OUT/kernel_stats.csv
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/pb_rd -o b -- python3 $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 2 --no-cpu-baseline --no-train-step --no-free-running > $OUT/bench_rd.log 2>&1
python3 $GRAFT_REPO_ROOT/tools/pmc_summary.py /tmp/pb_rd vnl_ > $OUT/pmc_traffic.txt
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d /tmp/pb_wr -o b -- python3 $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 2 --no-cpu-baseline --no-train-step --no-free-running > $OUT/bench_wr.log 2>&1
python3 $GRAFT_REPO_ROOT/tools/pmc_summary.py /tmp/pb_wr vnl_ >> $OUT/pmc_traffic.txt
python3 $GRAFT_REPO_ROOT/tools/pmc_traffic.py /tmp/pb_rd /tmp/pb_wr 4096 "${PROFILE_LABEL:-profiles/r03_pmc_traffic.txt}"
tail -1 $OUT/bench_stats.log | cut -c1-300
head -8 $OUT/kernel_stats.csv | cut -c1-160
cat $OUT/pmc_traffic.txt
