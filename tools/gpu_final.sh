#!/bin/bash
# final validation + profile refresh: everything that gets committed under profiles/ is produced here
set -u
TAG=${TAG:-r6}
REPO=$PWD
OUT=$REPO/gpurun_out
mkdir -p "$OUT"
export TMPDIR=/tmp
T0=$(date +%s)
# SKIP_TESTS=1: profile passes and bench lines only (a second box for the same, already validated tree)
if [ "${SKIP_TESTS:-0}" != "1" ]; then
echo "== GPU suite"
timeout 1500 python -m pytest tests -m gpu -q --timeout 600 -p no:cacheprovider -x > "$OUT/final_tests.log" 2>&1
tail -6 "$OUT/final_tests.log" | cut -c1-220
echo "== smoke ($(( $(date +%s) - T0 )) s)"
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
fi
echo "== profile passes first (the bench line then finds a PMC summary stamped with this tree) ($(( $(date +%s) - T0 )) s)"
bash tools/profile_bench.sh cfg2 $TAG > "$OUT/final_profile.log" 2>&1
python tools/pmc_sq_summary.py "$OUT/${TAG}_pmc_sq1_cfg2.csv" "$OUT/${TAG}_pmc_sq2_cfg2.csv" > "$OUT/${TAG}_pmc_sq_summary_cfg2.json" 2>/dev/null
# the reduced-precision modes: kernel trace + FETCH / WRITE passes (round 5: they had `traffic: null`)
SQ_PASSES=0 BENCH_ARGS="--mlp-precision bf16x3" bash tools/profile_bench.sh cfg2 $TAG _bf16x3 >> "$OUT/final_profile.log" 2>&1
SQ_PASSES=0 bash tools/profile_bench.sh cfg5 $TAG >> "$OUT/final_profile.log" 2>&1
# round 6: one rank's share of cfg3 at 8 ranks (2048-row minibatches: step16_kernel + dw_multi_kernel + fold), with the SQ passes
bash tools/profile_bench.sh cfg3_shard $TAG >> "$OUT/final_profile.log" 2>&1
python tools/pmc_sq_summary.py "$OUT/${TAG}_pmc_sq1_cfg3_shard.csv" "$OUT/${TAG}_pmc_sq2_cfg3_shard.csv" > "$OUT/${TAG}_pmc_sq_summary_cfg3_shard.json" 2>/dev/null
# the bench lines below look the PMC summaries up under profiles/ (stamped with a hash of csrc/): put this call's there first
mkdir -p profiles && cp "$OUT/${TAG}_pmc_traffic_cfg2.json" "$OUT/${TAG}_bench_cfg2_kernel_stats.csv" "$OUT/${TAG}_pmc_traffic_cfg2_bf16x3.json" \
   "$OUT/${TAG}_bench_cfg2_bf16x3_kernel_stats.csv" "$OUT/${TAG}_pmc_traffic_cfg5.json" "$OUT/${TAG}_bench_cfg5_kernel_stats.csv" \
   "$OUT/${TAG}_pmc_traffic_cfg3_shard.json" "$OUT/${TAG}_bench_cfg3_shard_kernel_stats.csv" profiles/ 2>/dev/null
bash tools/gpu_trace_one.sh reference $TAG > /dev/null 2>&1
python tools/explain_plan.py --all > "$OUT/${TAG}_explain_plan.txt" 2>/dev/null
echo "== bench lines ($(( $(date +%s) - T0 )) s)"
timeout 400 python bench.py 2> "$OUT/${TAG}_bench_cfg2.err" | tail -1 > "$OUT/${TAG}_bench_cfg2.json"
B="timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline"
$B --workload reference 2>/dev/null | tail -1 > "$OUT/${TAG}_bench_reference.json"
$B --workload cfg2 --mlp-precision bf16x3 2>/dev/null | tail -1 > "$OUT/${TAG}_bench_cfg2_bf16x3.json"
: > "$OUT/${TAG}_bench_other_configs.jsonl"
for WL in cfg1 cfg3 cfg3_shard cfg4 cfg5 cfg5_envs; do $B --workload $WL 2>/dev/null | tail -1 >> "$OUT/${TAG}_bench_other_configs.jsonl"; done
CATPPO_FORCE_DIST=1 $B --workload cfg2 2>/dev/null | tail -1 > "$OUT/${TAG}_bench_cfg2_forced_dist_world1.json"
: > "$OUT/${TAG}_bench_cfg3_shares.jsonl"
for W in 1 2 4 8; do $B --workload cfg3 --shard-of $W 2>/dev/null | tail -1 >> "$OUT/${TAG}_bench_cfg3_shares.jsonl"; done
echo "== launcher proof: bench.py starts two ranks itself (one GPU: gloo host staging) ($(( $(date +%s) - T0 )) s)"
timeout 600 python bench.py --gpus 2 --steps 5 --warmup 2 --no-cpu-baseline 2> "$OUT/${TAG}_bench_cfg2_gpus2_selflaunch.err" | tail -1 > "$OUT/${TAG}_bench_cfg2_gpus2_selflaunch.json"
echo "== three more default lines back to back (run-to-run spread) ($(( $(date +%s) - T0 )) s)"
: > "$OUT/${TAG}_bench_cfg2_repeats.jsonl"
for i in 1 2 3; do timeout 300 python bench.py --no-cpu-baseline 2>/dev/null | tail -1 >> "$OUT/${TAG}_bench_cfg2_repeats.jsonl"; done
if [ -f tools/bin/libcatppo_s16tl.so ]; then
  echo "== step16 timeline ($(( $(date +%s) - T0 )) s)"
  (for net in ref cfg2; do echo "==== $net"; CATPPO_LIB=$PWD/tools/bin/libcatppo_s16tl.so timeout 120 python tools/step16_timeline.py 2048 $net; done) 2>&1 | grep -v amdgpu.ids > "$OUT/${TAG}_step16_timeline.txt"
fi
if [ "${SKIP_TESTS:-0}" != "1" ] && [ -f tools/bin/libcatppo_fftl.so ]; then
  echo "== rows_fwd timeline ($(( $(date +%s) - T0 )) s)"
  (CATPPO_LIB=$PWD/tools/bin/libcatppo_fftl.so timeout 120 python tools/rows_fwd_timeline.py 16384 48 2; CATPPO_LIB=$PWD/tools/bin/libcatppo_fftl.so timeout 120 python tools/rows_fwd_timeline.py 16384 240 2) 2>&1 | grep -v amdgpu.ids > "$OUT/${TAG}_rows_fwd_timeline.txt"
fi
if [ "${SKIP_TESTS:-0}" != "1" ] && [ -f tools/bin/libcatppo_tl.so ]; then
  echo "== rollout timeline ($(( $(date +%s) - T0 )) s)"
  CATPPO_LIB=$PWD/tools/bin/libcatppo_tl.so timeout 200 python tools/rollout_timeline.py cfg2 2>&1 | grep -v "amdgpu.ids\|^\[INFO\]\|^Index\|^[0-9] |\|Active Constraint\|^$" > "$OUT/${TAG}_rollout_timeline_final.txt"
fi
if [ "${SKIP_TESTS:-0}" != "1" ] && [ -f tools/bin/libcatppo_r5d.so ]; then
  echo "== env step: this tree against the build with only the deferred tail ($(( $(date +%s) - T0 )) s)"
  ROUNDS=3 OUT=$OUT/${TAG}_ab_rollout_trim.jsonl bash tools/gpu_exp.sh "cfg2 deferred_tail_only CATPPO_LIB=$PWD/tools/bin/libcatppo_r5d.so" "cfg2 final X=1" > "$OUT/${TAG}_ab_rollout_trim.txt" 2>&1
  tail -3 "$OUT/${TAG}_ab_rollout_trim.txt"
fi
python - "$OUT" "$TAG" <<'PY'
import json,sys,os
out=sys.argv[1]; TAG=sys.argv[2]
def show(tag,line):
    try:
        d=json.loads(line)
        print(tag, round(d["value"]/1e6,3),"M/s ms",round(d["ms_per_step"],2),"grp_us",round(d["roofline"]["avg_launch_us"],1),"frac",round(d["roofline"]["frac"],3),"frac_profiled",d["roofline"].get("frac_profiled"),"traffic",d["roofline"]["traffic"],{k:round(v,2) for k,v in d["phases_device_ms"].items() if k!="iterations"})
        if d.get("secondary"):
            s2=d["secondary"]; print("   secondary", s2["dtype"][:7], round(s2["value"]/1e6,3),"M/s ms",round(s2["ms_per_step"],2),"grp_us",round(s2["roofline"]["avg_launch_us"],1),"frac",round(s2["roofline"]["frac"],3),"traffic",s2["roofline"]["traffic"])
    except Exception as e: print(tag,"FAILED",e)
for f in tuple(TAG + x for x in ("_bench_cfg2.json","_bench_reference.json","_bench_cfg2_bf16x3.json","_bench_cfg2_forced_dist_world1.json")):
    show(f, open(os.path.join(out,f)).read().strip().splitlines()[-1])
for f in (TAG + "_bench_cfg2_gpus2_selflaunch.json",):
    try:
        d=json.loads(open(os.path.join(out,f)).read().strip().splitlines()[-1])
        print(f, "n_gpus", d["n_gpus"], "physical", d["physical_gpus"], round(d["value"]/1e6,3), "M/s comm_ms", d["comm_ms_per_iteration"], d["config"]["collectives"][:40])
    except Exception as e: print(f, "FAILED", e)
for l in open(os.path.join(out,TAG + "_bench_cfg2_repeats.jsonl")):
    if l.strip(): show("repeat", l)
for l in open(os.path.join(out,TAG + "_bench_other_configs.jsonl")):
    if l.strip(): show(json.loads(l)["config"]["workload"][:12], l)
for l in open(os.path.join(out,TAG + "_bench_cfg3_shares.jsonl")):
    if l.strip(): show("cfg3 share of W=%s" % json.loads(l)["config"]["simulated_shard_of_world"], l)
PY
cat "$OUT/${TAG}_bench_cfg2.err" | grep bench
echo "== done ($(( $(date +%s) - T0 )) s)"
