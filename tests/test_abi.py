"""The C-ABI library loads and exports every symbol include/catppo.h declares (no GPU needed)."""
import ctypes
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    src = open(os.path.join(ROOT, "include", "catppo.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(catppo_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    import __graft_entry__ as ge
    ge.build()
    from cat_envs import native
    lib = ctypes.CDLL(native.LIB_PATH)
    names = _declared()
    assert len(names) >= 20
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/catppo.h but not exported"
    assert set(names) == set(native.EXPORTS), set(names) ^ set(native.EXPORTS)
    assert lib.catppo_version() == 600
    assert len(names) < 60, len(names)                  # ABI 0.6: the _ex families collapsed (77 exports in 0.5)


def test_layout_matches_reference_parameter_count():
    from cat_envs import native
    native.load_library()
    lay = native.layout_of(native.shape_of(45, 12, (512, 256, 128)))
    assert lay.n_params == 377241          # reference Agent
    assert lay.obs_pad == 48 and lay.n_flat % 4 == 0
    assert all(lay.off_w[net][l] % 32 == 0 for net in range(2) for l in range(4))      # weight matrices on 128-byte lines (round 6)
    lay2 = native.layout_of(native.shape_of(48, 12, (256, 256, 256)))
    assert lay2.obs_pad == 48 and lay2.n_params == 2 * (48 * 256 + 256 + 2 * (256 * 256 + 256)) + 256 * 13 + 13 + 12


def test_no_cpu_fallback():
    import pytest
    import torch
    from cat_envs import native
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        native.Native()
