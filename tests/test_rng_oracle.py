"""The numpy restatement of the on-device randomness (oracle/rng_oracle.py) is pinned here, on CPU: Philox4x32-10
against the Random123 known-answer vectors, Box-Muller moments, the keyed permutation's bijectivity.  The GPU tests
(tests/test_gpu_r2_features.py) then compare the kernels with this restatement element for element."""
import numpy as np
import pytest

from oracle import rng_oracle as R


def test_philox_known_answer_vectors():
    for ctr, key, exp in R.KAT:
        got = R.philox4x32_10(np.array(ctr, np.uint32), np.array(key, np.uint32))
        assert [int(x) for x in got] == list(exp)


def test_philox_is_vectorised_and_counter_sensitive():
    ctr = np.zeros((5, 3, 4), np.uint32)
    ctr[..., 0] = np.arange(5)[:, None]
    ctr[..., 1] = np.arange(3)[None, :]
    key = np.zeros((5, 3, 2), np.uint32)
    out = R.philox4x32_10(ctr, key)
    assert out.shape == (5, 3, 4) and len({tuple(r) for r in out.reshape(-1, 4).tolist()}) == 15
    np.testing.assert_array_equal(out[0, 0], R.philox4x32_10(np.zeros(4, np.uint32), np.zeros(2, np.uint32)))


def test_action_noise_moments_and_streams():
    z = R.action_noise(0xABCDEF0123456789, 3, 7, 50000, 12)
    assert z.shape == (50000, 12) and z.dtype == np.float32 and np.isfinite(z).all()
    assert abs(z.mean()) < 0.01 and abs(z.std() - 1) < 0.01 and abs((z ** 4).mean() - 3) < 0.1
    assert abs(np.corrcoef(z[:, 0], z[:, 1])[0, 1]) < 0.02          # the two outputs of one Box-Muller pair
    z2 = R.action_noise(0xABCDEF0123456789, 3, 8, 50000, 12)         # next rollout step: another stream
    z3 = R.action_noise(0xABCDEF0123456789, 4, 7, 50000, 12)         # next iteration: another stream
    assert abs(np.corrcoef(z.ravel(), z2.ravel())[0, 1]) < 0.01 and abs(np.corrcoef(z.ravel(), z3.ravel())[0, 1]) < 0.01
    np.testing.assert_array_equal(R.action_noise(5, 1, 0, 10, 7), R.action_noise(5, 1, 0, 10, 8)[:, :7])


@pytest.mark.parametrize("n", [1, 2, 35, 1000, 4096, 98304])
def test_keyed_permutation_is_a_bijection(n):
    p = R.permutation(42, 2, 1, n)
    assert p.dtype == np.int64 and sorted(p.tolist()) == list(range(n))
    if n >= 1000:
        q = R.permutation(42, 2, 2, n)                                # another epoch: another permutation
        assert (p != q).mean() > 0.99
        assert abs(np.corrcoef(np.arange(n), p)[0, 1]) < 4.0 / np.sqrt(n)  # no visible order left (sigma = 1/sqrt(n))
        # minibatch composition: each slice of the permutation is spread over the whole index range
        m = p[: n // 6]
        assert m.min() < n * 0.01 + 8 and m.max() > n * 0.99 - 8
