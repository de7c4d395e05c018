"""Round-6 GPU tests: the small-minibatch optimiser step (csrc/step16.h + dw_multi_kernel) against the layer-wise launches
it replaces (cleanrl/ppo.py:300-352 at the 2048-row minibatches of an env-sharded rank).  Everything goes through the C ABI."""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def _grad_of(tmp_path, tag, D, A, hidden, Bsz, M, **env_over):
    import test_gpu_kernels as TK
    out = str(tmp_path / f"{tag}.npz")
    code = TK._FUSED_VS_SPLIT.format(root=ROOT, D=D, A=A, hidden=hidden, Bsz=Bsz, M=M, prec=0, out=out)
    code = code.replace("nat.mlp_reserve(shape, M)", "nat.mlp_reserve(shape, M); nat.plan_log(1)")
    code = code.replace("np.savez(", "open({!r}, 'w').write(nat.plan_log(-1)); np.savez(".format(out + ".plan"))
    r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, **env_over), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    return np.load(out), open(out + ".plan").read()


@pytest.mark.parametrize("D,A,hidden,Bsz,M", [
    (45, 12, (512, 256, 128), 2048, 2048),        # one rank's minibatch of BASELINE configs[2] at 8 GPUs
    (48, 12, (256, 256, 256), 4096, 4096),        # cfg2's network at the top of the window: two tiles per CU
    (45, 12, (512, 256, 128), 1000, 300),         # ragged: 18 full tiles + 12 rows
    (48, 7, (256, 256, 256), 64, 17),             # two tiles, the second with one row
    (48, 12, (512, 256, 128), 512, 512),          # cfg1's minibatch
])
def test_small_minibatch_step_equals_the_layerwise_launches(tmp_path, D, A, hidden, Bsz, M):
    """step16_kernel + dw_multi_kernel (3 launches) against layer-wise forward GEMMs + fwd_head_kernel + paired dW / dX
    launches + dw_fold (10 launches) on the same minibatch.  The 16-row path walks every contraction in gemm_body's order
    on v_mfma_f32_16x16x4_f32 (one fp32 FMA chain per element, tools/mfma16_probe.hip), so the hidden layers' weight and
    bias gradients must be BIT-identical; the head partials cover 16 rows instead of 64 (another fold order): 2e-6 relative
    to the largest entry, like the diagnostics (sums of per-tile sums)."""
    from cat_envs import native
    shape = native.shape_of(D, A, hidden)
    lay = native.layout_of(shape)
    new, plan_new = _grad_of(tmp_path, "s16", D, A, hidden, Bsz, M, CATPPO_STEP16="1")
    old, plan_old = _grad_of(tmp_path, "lw", D, A, hidden, Bsz, M, CATPPO_STEP16="0", CATPPO_FUSED_HEAD_MIN_WG="1")
    assert "step16_kernel" in plan_new and "dw_multi_kernel" in plan_new and "gemm_pair_kernel" not in plan_new
    assert "step16_kernel" not in plan_old and "fwd_head_kernel" in plan_old and "gemm_pair_kernel" in plan_old
    gn, go = new["grad"], old["grad"]
    assert np.isfinite(gn).all() and np.abs(go).max() > 0
    nl = len(hidden)
    for net in range(2):
        for l in range(nl):
            w0, b0 = lay.off_w[net][l], lay.off_b[net][l]
            nxt = lay.off_w[net][l + 1]
            np.testing.assert_array_equal(gn[w0:nxt], go[w0:nxt], err_msg=f"net {net} hidden layer {l} (weights | bias)")
            assert np.abs(go[w0:b0]).max() > 0
    scale = float(np.abs(go).max())
    np.testing.assert_allclose(gn, go, rtol=0, atol=2e-6 * scale)                     # heads, log-std
    np.testing.assert_allclose(new["diag"], old["diag"], rtol=2e-6, atol=1e-7)


def test_small_minibatch_step_against_the_default_small_path(tmp_path):
    """the same step against what a 2048-row minibatch took through round 5 (layer-wise forward + head_loss_kernel:
    wave-per-row head sums) - different head order, so the whole gradient is compared at 3e-6 of its largest entry."""
    new, _ = _grad_of(tmp_path, "s16", 45, 12, (512, 256, 128), 2048, 2048, CATPPO_STEP16="1")
    old, plan_old = _grad_of(tmp_path, "r5", 45, 12, (512, 256, 128), 2048, 2048, CATPPO_STEP16="0")
    assert "head_loss_kernel" in plan_old
    scale = float(np.abs(old["grad"]).max())
    np.testing.assert_allclose(new["grad"], old["grad"], rtol=0, atol=3e-6 * scale)
    np.testing.assert_allclose(new["diag"], old["diag"], rtol=3e-6, atol=1e-7)


def test_16_row_rollout_forward_equals_the_layerwise_path(tmp_path):
    """step16_fwd_kernel (16-row tiles, three layers + rollout heads in one launch; cleanrl/ppo.py:104-119) against the
    layer-wise rollout forward at the batch sizes below the 32-row kernels' window: hidden layers bit-identical by
    construction, heads sum in another order (2e-6), Philox noise exact; ragged and one-row batches, critic-only call."""
    import test_gpu_kernels as TK
    cases = {"shard": (45, 12, (512, 256, 128), 2048), "ragged": (45, 12, (512, 256, 128), 1001), "tiny": (45, 12, (512, 256, 128), 33),
             "cfg1": (48, 12, (512, 256, 128), 64), "cfg2net": (48, 12, (256, 256, 256), 2048), "one_row": (48, 7, (256, 256, 256), 1)}
    outs = []
    for env_over in (dict(CATPPO_STEP16_FWD="1"), dict(CATPPO_STEP16_FWD="0", CATPPO_FUSED_FWD="0", CATPPO_ROWS_FWD_ROLLOUT="0", CATPPO_ROWS_WIDE="0")):
        out = str(tmp_path / f"s16f{len(outs)}.npz")
        code = TK._FUSED_FWD_AB.format(root=ROOT, cases=cases, out=out)
        code = code.replace("np.savez(", "nat.plan_log(1); nat.value(shape, params, T.dev(x), B, v2); open({!r}, 'w').write(nat.plan_log(-1)); np.savez(".format(out + ".plan"))
        r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, **env_over), capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-3000:]
        outs.append((np.load(out), open(out + ".plan").read()))
    (f, pf), (l, pl) = outs
    assert "step16_fwd_kernel" in pf and "step16_fwd_kernel" not in pl and "layer-wise" in pl
    for k in f.files:
        if k.endswith("_e3"):
            np.testing.assert_array_equal(f[k], l[k], err_msg=k)
        else:
            np.testing.assert_allclose(f[k], l[k], rtol=0, atol=2e-6 * max(1.0, float(np.abs(l[k]).max())), err_msg=k)
    assert np.abs(f["shard_act"]).max() > 0 and np.isfinite(f["ragged_lp"]).all()
