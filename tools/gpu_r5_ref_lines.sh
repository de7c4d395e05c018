set -u
export TMPDIR=/tmp
mkdir -p gpurun_out
: > gpurun_out/r5b_bench_reference_repeats.jsonl
for i in 1 2 3; do timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --workload reference 2>/dev/null | tail -1 >> gpurun_out/r5b_bench_reference_repeats.jsonl; done
python - <<'PY'
import json
for l in open("gpurun_out/r5b_bench_reference_repeats.jsonl"):
    d = json.loads(l); print("reference", round(d["value"]/1e6, 3), round(d["ms_per_step"], 2), round(d["roofline"]["avg_launch_us"], 1), {k: round(v, 2) for k, v in d["phases_device_ms"].items() if k != "iterations"})
PY
