#!/bin/bash
# round 5, first call: per-kernel traces of the reduced-precision modes (no per-kernel record existed) + a default line
set -u
export TMPDIR=/tmp
mkdir -p gpurun_out
BENCH_ARGS="--mlp-precision bf16x3" bash tools/gpu_trace_one.sh cfg2 r5base_bf16x3 > gpurun_out/r5base_bf16x3.txt 2>&1
mv gpurun_out/r5base_bf16x3_bench_cfg2_kernel_stats.csv gpurun_out/r5base_bench_cfg2_bf16x3_kernel_stats.csv
bash tools/gpu_trace_one.sh cfg5 r5base > gpurun_out/r5base_cfg5.txt 2>&1
timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/r5base_bench_cfg2.json
timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --mlp-precision bf16x3 2>/dev/null | tail -1 > gpurun_out/r5base_bench_cfg2_bf16x3.json
timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --workload cfg5 2>/dev/null | tail -1 > gpurun_out/r5base_bench_cfg5.json
python - <<'PY'
import json
for f in ("r5base_bench_cfg2","r5base_bench_cfg2_bf16x3","r5base_bench_cfg5"):
    try:
        d=json.loads(open("gpurun_out/%s.json"%f).read().strip().splitlines()[-1])
        print(f, round(d["value"]/1e6,3), "ms", round(d["ms_per_step"],2), "grp", round(d["roofline"]["avg_launch_us"],1), d["phases_device_ms"])
    except Exception as e: print(f,"FAILED",e)
PY
cut -c1-140 gpurun_out/r5base_bench_cfg2_bf16x3_kernel_stats.csv | head -12
cut -c1-140 gpurun_out/r5base_bench_cfg5_kernel_stats.csv | head -12
