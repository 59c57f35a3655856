timeout -k 10 300 python bench.py --no-train-step --no-cpu-baseline --no-parity > gpurun_out/q_bench.log 2>&1; python - <<PY
import json
for l in open("gpurun_out/q_bench.log"):
    if l.startswith("{"):
        j=json.loads(l); print("BENCH", j["value"], j["ms_per_step"], j["roofline"]["kernel_ms"], j["free_running"]["value"], j["free_running"]["kernel_ms"])
PY
