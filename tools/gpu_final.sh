#!/bin/bash
# final validation + profile refresh: everything that gets committed under profiles/ is produced here
set -u
REPO=$PWD
OUT=$REPO/gpurun_out
mkdir -p "$OUT"
export TMPDIR=/tmp
T0=$(date +%s)
echo "== GPU suite"
timeout 1500 python -m pytest tests -m gpu -q --timeout 600 -p no:cacheprovider -x > "$OUT/final_tests.log" 2>&1
tail -6 "$OUT/final_tests.log" | cut -c1-220
echo "== smoke ($(( $(date +%s) - T0 )) s)"
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
echo "== profile passes first (the bench line then finds a PMC summary stamped with this tree) ($(( $(date +%s) - T0 )) s)"
bash tools/profile_bench.sh cfg2 r3 > "$OUT/final_profile.log" 2>&1
python tools/pmc_sq_summary.py "$OUT/r3_pmc_sq1_cfg2.csv" "$OUT/r3_pmc_sq2_cfg2.csv" > "$OUT/r3_pmc_sq_summary_cfg2.json" 2>/dev/null
mkdir -p profiles && cp "$OUT/r3_pmc_traffic_cfg2.json" "$OUT/r3_bench_cfg2_kernel_stats.csv" profiles/ 2>/dev/null
echo "== bench lines ($(( $(date +%s) - T0 )) s)"
timeout 400 python bench.py 2> "$OUT/r3_bench_cfg2.err" | tail -1 > "$OUT/r3_bench_cfg2.json"
B="timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline"
$B --workload reference 2>/dev/null | tail -1 > "$OUT/r3_bench_reference.json"
$B --workload cfg2 --mlp-precision bf16x3 2>/dev/null | tail -1 > "$OUT/r3_bench_cfg2_bf16x3.json"
: > "$OUT/r3_bench_other_configs.jsonl"
for WL in cfg1 cfg3 cfg3_shard cfg4 cfg5 cfg5_envs; do $B --workload $WL 2>/dev/null | tail -1 >> "$OUT/r3_bench_other_configs.jsonl"; done
CATPPO_FORCE_DIST=1 $B --workload cfg2 2>/dev/null | tail -1 > "$OUT/r3_bench_cfg2_forced_dist_world1.json"
: > "$OUT/r3_bench_cfg3_shares.jsonl"
for W in 1 2 4 8; do $B --workload cfg3 --shard-of $W 2>/dev/null | tail -1 >> "$OUT/r3_bench_cfg3_shares.jsonl"; done
python - "$OUT" <<'PY'
import json,sys,os
out=sys.argv[1]
def show(tag,line):
    try:
        d=json.loads(line)
        print(tag, round(d["value"]/1e6,3),"M/s ms",round(d["ms_per_step"],2),"grp_us",round(d["roofline"]["avg_launch_us"],1),"frac",round(d["roofline"]["frac"],3),"traffic",d["roofline"]["traffic"],{k:round(v,2) for k,v in d["phases_device_ms"].items() if k!="iterations"})
    except Exception as e: print(tag,"FAILED",e)
for f in ("r3_bench_cfg2.json","r3_bench_reference.json","r3_bench_cfg2_bf16x3.json","r3_bench_cfg2_forced_dist_world1.json"):
    show(f, open(os.path.join(out,f)).read().strip().splitlines()[-1])
for l in open(os.path.join(out,"r3_bench_other_configs.jsonl")):
    if l.strip(): show(json.loads(l)["config"]["workload"][:12], l)
for l in open(os.path.join(out,"r3_bench_cfg3_shares.jsonl")):
    if l.strip(): show("cfg3 share of W=%s" % json.loads(l)["config"]["simulated_shard_of_world"], l)
PY
cat "$OUT/r3_bench_cfg2.err" | grep bench
echo "== done ($(( $(date +%s) - T0 )) s)"
