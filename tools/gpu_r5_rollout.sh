#!/bin/bash
# round 5, VERDICT r4 items 7 and 5: (a) rollout_post without its tail (catppo_rollout_defer_tail) and with its loads
# requested up front, (b) the no-extra-launch form of the gradient all-reduce overlap - full GPU suite, A/Bs, traces
set -u
export TMPDIR=/tmp
mkdir -p gpurun_out
T0=$(date +%s)
timeout 1500 python -m pytest tests -m gpu -q --timeout 600 -p no:cacheprovider -x > gpurun_out/rollout_tests.log 2>&1
tail -15 gpurun_out/rollout_tests.log | cut -c1-250
echo "== A/B deferred tail ($(( $(date +%s) - T0 )) s)"
ROUNDS=3 OUT=gpurun_out/r5_ab_rollout_defer_tail.jsonl bash tools/gpu_exp.sh \
  "cfg2 tail_in_post CATPPO_ROLLOUT_DEFER_TAIL=0" "cfg2 tail_deferred X=1" \
  "reference tail_in_post CATPPO_ROLLOUT_DEFER_TAIL=0" "reference tail_deferred X=1" > gpurun_out/r5_ab_rollout_defer_tail.txt 2>&1
tail -5 gpurun_out/r5_ab_rollout_defer_tail.txt
echo "== traces ($(( $(date +%s) - T0 )) s)"
CATPPO_ROLLOUT_DEFER_TAIL=0 bash tools/gpu_trace_one.sh cfg2 r5nodefer > /dev/null 2>&1
bash tools/gpu_trace_one.sh cfg2 r5defer > /dev/null 2>&1
python - <<'PY'
import csv
for tag in ("r5nodefer", "r5defer"):
    print(tag)
    try:
        for r in csv.DictReader(open(f"gpurun_out/{tag}_bench_cfg2_kernel_stats.csv")):
            k = r["kernel"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:60]
            if "rollout" in k or "rows_fwd_kernel<32" in k:
                print("  %-62s %-10s calls %5s avg %8s min %8s vgpr %s" % (k, r["blocks"], r["calls"], r["avg_us"], r["min_us"], r["vgpr"]))
    except Exception as e:
        print("  FAILED", e)
PY
echo "== A/B gradient all-reduce overlap, world of one ($(( $(date +%s) - T0 )) s)"
F="CATPPO_FORCE_DIST=1"
ROUNDS=2 OUT=gpurun_out/r5_grad_overlap_tail_world1.jsonl bash tools/gpu_exp.sh \
  "cfg2 graph_one_allreduce $F" "cfg2 graph_per_layer $F CATPPO_GRAD_OVERLAP=1" "cfg2 graph_tail $F CATPPO_GRAD_OVERLAP=2" \
  "cfg2 eager_one_allreduce $F CATPPO_GRAPH_UPDATE=0" "cfg2 eager_per_layer $F CATPPO_GRAPH_UPDATE=0 CATPPO_GRAD_OVERLAP=1" "cfg2 eager_tail $F CATPPO_GRAPH_UPDATE=0 CATPPO_GRAD_OVERLAP=2" \
  > gpurun_out/r5_grad_overlap_tail_world1.txt 2>&1
tail -7 gpurun_out/r5_grad_overlap_tail_world1.txt
echo "== done ($(( $(date +%s) - T0 )) s)"
