#!/bin/bash
# A/B of run-time switches inside ONE gpurun call (box-to-box variance is +-3-5 %: only same-call comparisons count).
#   bash tools/gpu_exp.sh "<workload> <tag> [VAR=value ...]" ...        each argument = one bench run; all of them are
#                                                                        repeated ROUNDS (default 3) times, INTERLEAVED
#                                                                        (A B A B ...: the driver's own record shows a
#                                                                        2.4 % order effect between two back-to-back
#                                                                        loops of the same work), and a summary with
#                                                                        mean / min / max / spread per tag is printed
#   EXTRA="--mlp-precision bf16x3" ...                                   extra bench.py arguments for every run
#   PYTEST="tests/test_gpu_kernels.py -k minibatch"                      optional parity run in front
#   OUT=gpurun_out/exp.jsonl                                             where the per-run records go
# example:  bash tools/gpu_exp.sh "cfg2 two_launches CATPPO_FUSED_HEAD=0" "cfg2 fused" "cfg5 two_launches CATPPO_FUSED_HEAD=0" "cfg5 fused"
set -u
mkdir -p gpurun_out
OUT=${OUT:-gpurun_out/exp.jsonl}
: > $OUT
[ -n "${PYTEST:-}" ] && timeout 1200 python -m pytest $PYTEST -m gpu -q -x -p no:cacheprovider 2>&1 | tail -3
SPECS=("$@")
for r in $(seq 1 ${ROUNDS:-3}); do
  for spec in "${SPECS[@]}"; do
    set -- $spec; wl=$1; tag=$2; shift 2
    env "$@" X=1 timeout 300 python bench.py --steps ${STEPS:-10} --warmup 3 --no-cpu-baseline --workload $wl ${EXTRA:-} 2>/dev/null | tail -1 | python -c "
import json,sys
try:
    d=json.loads(sys.stdin.read())
except Exception as e:
    print('%-11s %-28s FAILED' % ('$wl','$tag')); sys.exit(0)
rec={'wl':'$wl','tag':'$tag','round':$r,'M_per_s':d['value']/1e6,'grp_us':d['roofline']['avg_launch_us'],'update_ms':d['phases_device_ms']['update_ms'],'rollout_ms':d['phases_device_ms']['rollout_ms'],'ms':d['ms_per_step']}
open('$OUT','a').write(json.dumps(rec)+'\n')
print('%-11s %-28s' % ('$wl','$tag'), round(rec['M_per_s'],3),'M/s grp_us',round(rec['grp_us'],1),'update_ms',round(rec['update_ms'],3),'rollout_ms',round(rec['rollout_ms'],3))"
  done
done
python - <<PY
import json, collections
rows = [json.loads(l) for l in open("$OUT")]
by = collections.OrderedDict()
for r in rows:
    by.setdefault((r["wl"], r["tag"]), []).append(r)
print("---- summary (mean [min .. max] over the interleaved rounds; spread = (max - min) / mean)")
for (wl, tag), rs in by.items():
    def st(k):
        v = [r[k] for r in rs]; m = sum(v) / len(v)
        return "%8.3f [%8.3f .. %8.3f] %4.1f%%" % (m, min(v), max(v), 100 * (max(v) - min(v)) / m)
    print("%-11s %-28s n=%d  ms/iter %s | grp_us %s | update_ms %s | rollout_ms %s" % (wl, tag, len(rs), st("ms"), st("grp_us"), st("update_ms"), st("rollout_ms")))
PY
