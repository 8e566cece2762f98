#!/bin/bash
cd /root/repo
t0=$(date +%s)
python bench.py > gpurun_out/final_bench_r4.json 2> gpurun_out/final_bench_r4.err
echo "bench.py wall: $(( $(date +%s) - t0 )) s"
python3 - <<'PY'
import json
d=json.loads(open("gpurun_out/final_bench_r4.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["ms_per_step"], d["config"]["x3_overflow"], d["config"]["library_fallbacks"])
print("bf16", d["bf16_mode"]["value"], "train", d["train_step"]["value"], d["train_step"]["bf16_mode"]["value"], "cpu", d["cpu_baseline"]["value"])
for k,v in d["kernels"].items(): print(k, v.get("frac"), v.get("launch_ms"))
PY
