#!/bin/bash
# rocprofv3 passes over bench.py on the GPU box; summaries land in gpurun_out/ (copy the ones to keep into profiles/).
#   bash tools/profile_bench.sh [workload] [tag] [name suffix, e.g. _bf16x3 with BENCH_ARGS="--mlp-precision bf16x3"]
# ROCm 7.2's rocprofv3 writes a rocpd SQLite database and can hang at process exit: every pass runs under
# `timeout -s KILL` and is summarised from the database on the box (the databases are too big to bring back).
# Counter passes are separate runs with nothing but --pmc (gpurun refuses --pmc mixed with trace domains).
set -u
WL=${1:-cfg2}
TAG=${2:-r5}
SFX=${3:-}
OUT=$PWD/gpurun_out
mkdir -p "$OUT"
export TMPDIR=/tmp
CMD="python $PWD/bench.py --workload $WL --steps 3 --warmup 1 --no-cpu-baseline ${BENCH_ARGS:-}"
run_pass() {   # name, rocprof args...
  local name=$1; shift
  local dir=/tmp/prof_$name
  rm -rf "$dir"
  (cd /tmp && timeout -s KILL 240 rocprofv3 "$@" -d "$dir" -- $CMD > "$OUT/${TAG}_${name}.log" 2>&1)
  find "$dir" -name '*.db' | head -1
}
DB=$(run_pass trace --kernel-trace)
[ -n "$DB" ] && python tools/rocpd_stats.py "$DB" > "$OUT/${TAG}_bench_${WL}${SFX}_kernel_stats.csv"
for C in FETCH_SIZE WRITE_SIZE; do
  DB=$(run_pass "pmc_$C" --pmc "$C")
  [ -n "$DB" ] && python tools/rocpd_pmc.py "$DB" > "$OUT/${TAG}_pmc_${C}_${WL}${SFX}.csv"
done
# SQ passes (8 SQ slots each): where the waves of every kernel spend their cycles, MFMA busy time, LDS conflicts
if [ "${SQ_PASSES:-1}" = "1" ]; then
  DB=$(run_pass pmc_sq1 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY \
       SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_LDS_BANK_CONFLICT)
  [ -n "$DB" ] && python tools/rocpd_pmc.py "$DB" > "$OUT/${TAG}_pmc_sq1_${WL}${SFX}.csv"
  DB=$(run_pass pmc_sq2 --pmc SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_LDS_IDX_ACTIVE \
       SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_INSTS_VALU SQ_WAVES)
  [ -n "$DB" ] && python tools/rocpd_pmc.py "$DB" > "$OUT/${TAG}_pmc_sq2_${WL}${SFX}.csv"
fi
if [ -s "$OUT/${TAG}_pmc_FETCH_SIZE_${WL}${SFX}.csv" ] && [ -s "$OUT/${TAG}_pmc_WRITE_SIZE_${WL}${SFX}.csv" ]; then
  python tools/pmc_group_traffic.py "$OUT/${TAG}_pmc_FETCH_SIZE_${WL}${SFX}.csv" "$OUT/${TAG}_pmc_WRITE_SIZE_${WL}${SFX}.csv" \
    > "$OUT/${TAG}_pmc_traffic_${WL}${SFX}.json"
fi
ls -la "$OUT" | tail -12
