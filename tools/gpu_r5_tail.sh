#!/bin/bash
# VERDICT r4 item 5 measured: catppo_set_grad_overlap(ctx, 2) - the no-extra-launch "tail" form - against one all-reduce
# and against round 4's per-layer buckets, every exchange point forced on over a world of one; then the PMC / trace
# summaries of the three profiled workloads re-stamped for this tree.
set -u
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_r5.py tests/test_gpu_r4.py -m gpu -q -x -p no:cacheprovider -k "tail or gradient_buckets" 2>&1 | tail -3
F="CATPPO_FORCE_DIST=1"
ROUNDS=2 OUT=gpurun_out/r5_grad_overlap_tail_world1.jsonl bash tools/gpu_exp.sh \
  "cfg2 graph_one_allreduce $F" "cfg2 graph_per_layer $F CATPPO_GRAD_OVERLAP=1" "cfg2 graph_tail $F CATPPO_GRAD_OVERLAP=2" \
  "cfg2 eager_one_allreduce $F CATPPO_GRAPH_UPDATE=0" "cfg2 eager_per_layer $F CATPPO_GRAPH_UPDATE=0 CATPPO_GRAD_OVERLAP=1" "cfg2 eager_tail $F CATPPO_GRAPH_UPDATE=0 CATPPO_GRAD_OVERLAP=2" \
  > gpurun_out/r5_grad_overlap_tail_world1.txt 2>&1
tail -8 gpurun_out/r5_grad_overlap_tail_world1.txt
SQ_PASSES=0 bash tools/profile_bench.sh cfg2 r5 > gpurun_out/tail_profile.log 2>&1
SQ_PASSES=0 BENCH_ARGS="--mlp-precision bf16x3" bash tools/profile_bench.sh cfg2 r5 _bf16x3 >> gpurun_out/tail_profile.log 2>&1
SQ_PASSES=0 bash tools/profile_bench.sh cfg5 r5 >> gpurun_out/tail_profile.log 2>&1
mkdir -p profiles && cp gpurun_out/r5_pmc_traffic_cfg2.json gpurun_out/r5_pmc_traffic_cfg2_bf16x3.json gpurun_out/r5_pmc_traffic_cfg5.json \
  gpurun_out/r5_bench_cfg2_kernel_stats.csv gpurun_out/r5_bench_cfg2_bf16x3_kernel_stats.csv gpurun_out/r5_bench_cfg5_kernel_stats.csv profiles/ 2>/dev/null
timeout 400 python bench.py 2> gpurun_out/r5_bench_cfg2.err | tail -1 > gpurun_out/r5_bench_cfg2.json
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r5_bench_cfg2.json").read())
r = d["roofline"]
print("cfg2", round(d["value"] / 1e6, 3), "M/s grp_us", round(r["avg_launch_us"], 1), "frac", round(r["frac"], 3), "profiled", r.get("frac_profiled"), "traffic", r["traffic"], r.get("traffic_note"))
s = d.get("secondary")
if s: print("secondary", round(s["value"] / 1e6, 3), round(s["roofline"]["avg_launch_us"], 1), s["roofline"]["traffic"])
PY
