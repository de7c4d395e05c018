#!/bin/bash
# bench lines of the committed tree on another box (box-to-box spread is 2-4 %: profiles keep both)
set -u
export TMPDIR=/tmp
OUT=gpurun_out; mkdir -p $OUT
timeout 400 python bench.py 2> $OUT/r5b_bench_cfg2.err | tail -1 > $OUT/r5b_bench_cfg2.json
B="timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline"
: > $OUT/r5b_bench_cfg2_repeats.jsonl
for i in 1 2 3; do timeout 300 python bench.py --no-cpu-baseline 2>/dev/null | tail -1 >> $OUT/r5b_bench_cfg2_repeats.jsonl; done
: > $OUT/r5b_bench_other_configs.jsonl
for WL in reference cfg3 cfg3_shard cfg4 cfg5; do $B --workload $WL 2>/dev/null | tail -1 >> $OUT/r5b_bench_other_configs.jsonl; done
python - <<'PY'
import json
def show(tag, l):
    d = json.loads(l); r = d["roofline"]
    print(tag, round(d["value"]/1e6, 3), "M/s ms", round(d["ms_per_step"], 2), "grp_us", round(r["avg_launch_us"], 1), "frac", round(r["frac"], 3), "traffic", r["traffic"], {k: round(v, 2) for k, v in d["phases_device_ms"].items() if k != "iterations"})
    s = d.get("secondary")
    if s: print("   secondary", round(s["value"]/1e6, 3), "M/s", s["passes_ms_per_step"], "grp_us", round(s["roofline"]["avg_launch_us"], 1))
show("cfg2", open("gpurun_out/r5b_bench_cfg2.json").read())
for l in open("gpurun_out/r5b_bench_cfg2_repeats.jsonl"):
    if l.strip(): show("repeat", l)
for l in open("gpurun_out/r5b_bench_other_configs.jsonl"):
    if l.strip(): show(json.loads(l)["config"]["workload"][:12], l)
PY
