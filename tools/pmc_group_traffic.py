"""HBM traffic per catppo_ppo_minibatch_grad launch from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE)
summarised by tools/rocpd_pmc.py.  gfx950 correction per MI355X_MICROARCH.md (HBM section): FETCH_SIZE counts
128-B requests at 64 B for wide (16 B/lane) coalesced streams -> doubled; both counters are in KiB.
Calibration in the same run: gae_scan<1> at 4096x24 reads 4 planes + 3 rows = 1,622,016 B and writes 2 planes =
786,432 B; the counters give FETCH 810.5 KiB (x2 = 1,659,904 B, +2 %) and WRITE 768 KiB (= 786,432 B, exact).

python tools/pmc_group_traffic.py gpurun_out/pmc_FETCH_SIZE.csv gpurun_out/pmc_WRITE_SIZE.csv > profiles/rN_pmc_traffic.json"""
import csv
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

GROUP = ("gemm_f32_kernel", "gemm_pair_kernel", "head_loss_kernel", "fwd_head_kernel", "seg_reduce_kernel", "rows_fwd_kernel", "dw_fold_kernel", "rows_fwd_wide_kernel", "step16_kernel", "dw_multi_kernel", "fwd0_w16_kernel")


def per_launch(path):
    rows = list(csv.DictReader(open(path)))
    n_mb = next(int(r["calls"]) for r in rows if "seg_reduce_kernel" in r["kernel"])      # one fold per minibatch
    kib, detail = 0.0, {}
    for r in rows:
        if not any(k in r["kernel"] for k in GROUP):
            continue
        s, calls = float(r["sum"]), int(r["calls"])
        if "<64, 64, true, true, 0, 64" in r["kernel"] or "rows_fwd_kernel<32" in r["kernel"] or "rows_fwd_wide_kernel<32" in r["kernel"]:
            continue                    # 64-wide contraction slabs: the rollout's policy GEMMs only (<= 4096 rows)
        if r["kernel"].rstrip().split("(")[0].endswith(", 6>"):
            continue                    # bf16-stored input, fp32 output: the rollout forward's last hidden layer only
        if "<64, 64, true, true, 0" in r["kernel"]:
            s = s / calls * n_mb        # 64x64 forward launches also serve the rollout: one (K=Dp layer) per minibatch
        elif "fwd0_w16_kernel" in r["kernel"] or ("gemm_f32_kernel<" in r["kernel"] and "true, true, 0, 16" in r["kernel"]):
            # layer-wise forward launches of the bf16-operand mode serve the rollout too (a few calls per env step against
            # hundreds per update phase): whole launches per minibatch x the average call
            s = s / calls * max(1, round(calls / n_mb)) * n_mb
        kib += s
        detail[r["kernel"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][-70:]] = \
            round(float(r["avg"]), 1)
    gae = [float(r["avg"]) for r in rows if "gae_scan<1" in r["kernel"]]
    return kib / n_mb, n_mb, detail, (gae[0] if gae else None)


f, n, fd, fg = per_launch(sys.argv[1])
w, _, wd, wg = per_launch(sys.argv[2])
import bench  # noqa: E402  (csrc_hash: bench.py reports traffic only from a summary measured on the current kernels)

out = {"kernel_group": "catppo_ppo_minibatch_grad_packed", "minibatches_profiled": n, "csrc_hash": bench.csrc_hash(),
       "FETCH_SIZE_KiB_per_launch_raw": f, "WRITE_SIZE_KiB_per_launch": w,
       "hbm_bytes_per_launch": (2.0 * f + w) * 1024.0,
       "correction": "FETCH_SIZE x2 (gfx950, wide coalesced loads), WRITE_SIZE x1, KiB -> bytes",
       "gae_4096x24_calibration": {"FETCH_SIZE_KiB": fg, "WRITE_SIZE_KiB": wg, "algorithmic_read_bytes": 1622016,
                                    "algorithmic_write_bytes": 786432},
       "avg_KiB_per_kernel_call": {"FETCH_SIZE": fd, "WRITE_SIZE": wd}}
print(json.dumps(out, indent=1))
