#!/bin/bash
# per-kernel rocprofv3 traces of several library builds in one gpurun call:  tools/gpu_trace_variants.sh name ...
set -u
R=$PWD; OUT=$R/gpurun_out; mkdir -p "$OUT"; export TMPDIR=/tmp
WL=${WL:-cfg2}
for name in "$@"; do
  if [ "$name" = tree ]; then lib=$R/constraints-as-terminations_amd/lib/libcatppo.so; else lib=$R/tools/bin/libcatppo_$name.so; fi
  dir=/tmp/prof_$name; rm -rf $dir
  (cd /tmp && CATPPO_LIB=$lib timeout -s KILL 200 rocprofv3 --kernel-trace -d $dir -- python $R/bench.py --workload $WL --steps 3 --warmup 1 --no-cpu-baseline > "$OUT/trace_$name.log" 2>&1)
  DB=$(find $dir -name '*.db' | head -1)
  [ -n "$DB" ] && python tools/rocpd_stats.py "$DB" > "$OUT/kstats_${WL}_$name.csv"
  echo "== $name"; python - <<PY
import csv
rows=list(csv.DictReader(open("$OUT/kstats_${WL}_$name.csv")))
for r in rows:
    k=r['kernel']
    if any(x in k for x in ("gemm_pair","fwd_head","gemm_f32_kernel","seg_reduce","clip_adam","sqnorm","head_act")) and int(r['calls'])>=100:
        print("  %-78s %-9s %4s %7s" % (k.replace('(anonymous namespace)::','').replace('void ','')[:78], r['blocks'], r['calls'], r['avg_us']))
PY
done
