"""Phase timeline of the rollout forward (rows_fwd_kernel<32> with heads: one workgroup per (32-row tile, network)): wall-clock
stamps of thread 0 of every workgroup + shader-clock stamps around the last layer's contraction.  Needs -DFUSED_TL:
    python constraints-as-terminations_amd/build.py --variant fftl -DFUSED_TL
    CATPPO_LIB=$PWD/tools/bin/libcatppo_fftl.so python tools/rows32_rollout_timeline.py
stamp indices: 0 entry, 1 observation tile in LDS, 2 + 2 l contraction of layer l done (wave 0), 3 + 2 l its epilogue done"""
import ctypes as C, os, sys
import numpy as np, torch
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
for p in (ROOT, os.path.join(ROOT, "constraints-as-terminations_amd")):
    sys.path.insert(0, p)
from cat_envs import native
torch.cuda.set_device(0)
rows = 4096
nat = native.get(torch.device("cuda", 0))
shape = native.shape_of(48, 12, (256, 256, 256))
lay = native.layout_of(shape)
g = torch.Generator(device="cuda").manual_seed(1)
flat = torch.randn(lay.n_flat, device="cuda", generator=g) * 0.05
x = torch.randn(rows, lay.obs_pad, device="cuda", generator=g)
eps = torch.randn(rows, 12, device="cuda", generator=g)
act, lp, val = torch.empty(rows, 12, device="cuda"), torch.empty(rows, device="cuda"), torch.empty(rows, device="cuda")
nat.mlp_reserve(shape, rows)
lib = nat.lib
lib.catppo_debug_fused_tl.restype, lib.catppo_debug_fused_tl.argtypes = C.c_int, [C.c_void_p]
buf = torch.zeros(2 * 1024 * 16 + 2 * 1024 * 4, dtype=torch.int64, device="cuda")
for _ in range(5):
    nat.policy_act(shape, flat, x, rows, eps, act, lp, val)
torch.cuda.synchronize()
assert lib.catppo_debug_fused_tl(buf.data_ptr()) == 0
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); nat.policy_act(shape, flat, x, rows, eps, act, lp, val); e1.record(); torch.cuda.synchronize()
print("launch %.1f us" % (e0.elapsed_time(e1) * 1e3))
allraw = buf.cpu().numpy()
raw = allraw[:2 * 1024 * 16].reshape(2, 1024, 16).astype(np.float64)
ck = allraw[2 * 1024 * 16:].reshape(2, 1024, 4).astype(np.float64)
for net in (0, 1):
    c = ck[net]; c = c[c[:, 0] > 0]
    if len(c) == 0: continue
    dt_us, ticks = (c[:, 3] - c[:, 2]) * 0.01, c[:, 1] - c[:, 0]
    print("net %d last layer wave 0: %.2f us wall, %.0f ticks -> %.3f GHz; 128 MFMAs of this wave = %.1f ticks each" % (net, np.median(dt_us), np.median(ticks), np.median(ticks / dt_us) / 1e3, np.median(ticks) / 128))
    t = raw[net] * 0.01; wg = t[t[:, 0] > 0]; t0 = wg[:, 0].min()
    for i in range(16):
        v = wg[:, i]
        if (v > 0).all(): print("   stamp %2d p50 %6.2f" % (i, np.median(v) - t0))
