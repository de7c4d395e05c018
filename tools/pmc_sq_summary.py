"""Per-kernel SQ counter ratios from the two `--pmc` SQ passes of tools/profile_bench.sh (rocpd_pmc.py CSVs).

    python tools/pmc_sq_summary.py gpurun_out/r1_pmc_sq1_cfg2.csv gpurun_out/r1_pmc_sq2_cfg2.csv > profiles/r1_pmc_sq_summary_cfg2.json

wave-cycle split (MI355X guide, "rocprofv3 PMC slots"): WAIT_ANY = parked on s_waitcnt / barrier, WAIT_INST_ANY = ready
but not issued (MFMA pipe busy, dependent MFMA, ...), ACTIVE_INST_ANY = issuing; the three sum to ~WAVE_CYCLES.
mfma_util = SQ_VALU_MFMA_BUSY_CYCLES / (SIMDs per counter instance x SQ_BUSY_CYCLES): the counters are reported per
shader engine (32 instances per dispatch = 8 CUs = 32 SIMDs each), MFMA busy cycles are summed over the SIMDs."""
import collections
import csv
import json
import sys

SIMDS_PER_INSTANCE = 32
KEEP = ("gemm", "dw_fold", "rows_fwd", "fused_fwd", "head_loss", "fwd_head", "head_act", "seg_reduce", "cat_", "rms_", "gae_scan", "clip_adam", "sqnorm", "ppo_gather", "rollout_pre", "rollout_fold", "rollout_post",
        "env_pre_step", "rollout_store")


def load(path):
    d = collections.defaultdict(dict)
    for r in csv.DictReader(open(path)):
        d[r["kernel"]][r["counter"]] = (int(r["calls"]), float(r["sum"]))
    return d


a, b = load(sys.argv[1]), load(sys.argv[2])
out = {}
for k in sorted(a, key=lambda k: -a[k].get("SQ_WAVE_CYCLES", (0, 0))[1]):
    if not any(x in k for x in KEEP) or "SQ_WAVE_CYCLES" not in a[k]:
        continue
    A, B = a[k], b.get(k, {})
    wc = A["SQ_WAVE_CYCLES"][1]

    def frac(d, c, den=wc):
        return round(d[c][1] / den, 4) if c in d and den else None
    name = k.replace("(anonymous namespace)::", "").replace("void ", "")[:90]
    out[name] = {
        "dispatches": A["SQ_WAVE_CYCLES"][0] // 32,
        "wave_cycles_parked_waitcnt_barrier": frac(A, "SQ_WAIT_ANY"),
        "wave_cycles_issue_stalled": frac(A, "SQ_WAIT_INST_ANY"),
        "wave_cycles_issuing": frac(A, "SQ_ACTIVE_INST_ANY"),
        "mfma_util": round(A["SQ_VALU_MFMA_BUSY_CYCLES"][1] / (SIMDS_PER_INSTANCE * A["SQ_BUSY_CYCLES"][1]), 4)
        if A.get("SQ_BUSY_CYCLES", (0, 0))[1] else None,
        "mfma_f32_mops_per_dispatch": round(A["SQ_INSTS_VALU_MFMA_MOPS_F32"][1] / max(A["SQ_WAVE_CYCLES"][0] // 32, 1))
        if "SQ_INSTS_VALU_MFMA_MOPS_F32" in A else None,
        "lds_bank_conflict_cycles_per_lds_active_cycle": frac(A, "SQ_LDS_BANK_CONFLICT",
                                                              B.get("SQ_LDS_IDX_ACTIVE", (0, 0))[1]),
        "issue_lds": frac(B, "SQ_ACTIVE_INST_LDS"), "issue_valu": frac(B, "SQ_ACTIVE_INST_VALU"),
        "issue_vmem": frac(B, "SQ_ACTIVE_INST_VMEM"), "issue_stalled_on_lds": frac(B, "SQ_WAIT_INST_LDS"),
    }
print(json.dumps(out, indent=1))
