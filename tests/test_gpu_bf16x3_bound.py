"""The SPECIFIED error bound of the opt-in split-bf16 contraction mode (include/mi_rl.h: MI_BF16X3_REL_BOUND, MI_BF16X3_ABS_FLOOR) on adversarial
operands — not the learning distribution.  mi_test_contraction runs the gradient kernels' own building blocks (split8 / bx_mac on v_mfma_f32_16x16x32_bf16 for
MI_CONTRACTION_BF16X3, v_mfma_f32_16x16x4_f32 for MI_CONTRACTION_F32: the nn.Linear(64, 64) contractions of ppo.py:56-63 under ppo.py:190) on one 16 x K x 16 tile; the
exact result is the float64 product of the same f32 operands.

    | y - y_exact |  <=  REL * sum_k |a_k| |b_k|  +  K * ABS_FLOOR            REL = 2^-20 max(1, K / 64) for bf16x3;   2^-24 (K / 4 + 1) for the exact-f32 path (same form)

Cases: log-uniform magnitudes over 2^-20 .. 2^+20 per element with random signs; cancelling pairs (+x, -x (1 + 2^-12)) whose result is ~2^-12 of the absolute sum;
one huge term among tiny ones; subnormal operands and products; small integers (both modes must be EXACT)."""
import ctypes as C

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
REL_BX, ABS_FLOOR = 2.0 ** -20, 2.0 ** -126


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("needs an MI355X")
    return torch.device("cuda", 0)


def _run(dev, mode, A, B):
    from deep_rl_amd import _native as N

    K = A.shape[1]
    a, b = torch.from_numpy(A).to(dev), torch.from_numpy(B).to(dev)
    d = torch.zeros(16, 16, dtype=torch.float32, device=dev)
    N.check(N.lib().mi_test_contraction(mode, N.ptr(a), N.ptr(b), K, N.ptr(d), N.stream_ptr(dev)), "mi_test_contraction")
    return d.cpu().numpy().astype(np.float64)


def _cases(rng, K):
    sgn = lambda s: rng.choice([-1.0, 1.0], size=s)  # noqa: E731
    logu = lambda s, lo, hi: (2.0 ** rng.uniform(lo, hi, size=s) * sgn(s)).astype(np.float32)  # noqa: E731
    yield "magnitudes 2^-20..2^+20", logu((16, K), -20, 20), logu((K, 16), -20, 20)
    yield "unit scale", rng.standard_normal((16, K)).astype(np.float32), rng.standard_normal((K, 16)).astype(np.float32)
    # catastrophic cancellation: consecutive k carry (+x, -x (1 + 2^-12)) against equal b: the result is ~2^-12 of the absolute sum
    A = logu((16, K), -8, 8); A[:, 1::2] = -A[:, 0::2] * np.float32(1 + 2.0 ** -12)
    B = logu((K, 16), -8, 8); B[1::2, :] = B[0::2, :]
    yield "cancelling pairs", A, B
    A = logu((16, K), -20, -10); A[:, 7] = logu((16,), 15, 20)
    yield "one huge term among tiny ones", A, logu((K, 16), -3, 3)
    yield "subnormal operands", logu((16, K), -140, -127), logu((K, 16), -2, 2)
    yield "subnormal products", logu((16, K), -70, -60), logu((K, 16), -75, -62)
    yield "operands with all 24 mantissa bits set", (np.float32(2.0) - np.float32(2.0 ** -23)) * sgn((16, K)).astype(np.float32) * logu((16, K), -4, 4), \
        (np.float32(2.0) - np.float32(2.0 ** -23)) * np.ones((K, 16), np.float32) * logu((K, 16), -4, 4)


@pytest.mark.parametrize("K", [64, 32, 256])
def test_bf16x3_contraction_error_bound(dev, K):
    rng = np.random.default_rng(1000 + K)
    worst = {"bx": 0.0, "f32": 0.0}
    for rep in range(20):
        for name, A, B in _cases(rng, K):
            exact = A.astype(np.float64) @ B.astype(np.float64)
            absum = np.abs(A).astype(np.float64) @ np.abs(B).astype(np.float64)
            y_bx, y_f32 = _run(dev, 1, A, B), _run(dev, 0, A, B)
            assert np.isfinite(y_bx).all() and np.isfinite(y_f32).all(), name
            e_bx, e_f32 = np.abs(y_bx - exact), np.abs(y_f32 - exact)
            assert (e_bx <= REL_BX * max(1, K / 64) * absum + K * ABS_FLOOR).all(), (name, K, float((e_bx / (absum + 1e-300)).max()))
            assert (e_f32 <= 2.0 ** -24 * (K / 4 + 1) * absum + K * ABS_FLOOR).all(), (name, K, float((e_f32 / (absum + 1e-300)).max()))
            big = absum > 2.0 ** -100                     # where the relative part of the bound is the binding one
            if big.any():
                worst["bx"] = max(worst["bx"], float((e_bx[big] / absum[big]).max()))
                worst["f32"] = max(worst["f32"], float((e_f32[big] / absum[big]).max()))
    print("K = %d: worst |err| / sum|a||b|: bf16x3 2^%.2f (bound 2^%.2f), exact-f32 MFMA 2^%.2f (bound 2^%.2f)"
          % (K, np.log2(worst["bx"]), np.log2(REL_BX * max(1, K / 64)), np.log2(worst["f32"]), np.log2(2.0 ** -24 * (K / 4 + 1))))
    assert worst["bx"] <= REL_BX * max(1, K / 64)


def test_small_integers_are_exact_in_both_modes(dev):
    rng = np.random.default_rng(7)
    for K in (32, 64, 128):
        A = rng.integers(-200, 201, size=(16, K)).astype(np.float32)
        B = rng.integers(-200, 201, size=(K, 16)).astype(np.float32)
        exact = A.astype(np.float64) @ B.astype(np.float64)      # |sum| < 2^24: exact in f32 whatever the order
        assert np.array_equal(_run(dev, 1, A, B), exact) and np.array_equal(_run(dev, 0, A, B), exact)


def test_bound_constants_match_the_header():
    import os
    import re

    hdr = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "include", "mi_rl.h")).read()
    assert float(re.search(r"#define MI_BF16X3_REL_BOUND (\S+)", hdr).group(1)) == REL_BX
    assert float(re.search(r"#define MI_BF16X3_ABS_FLOOR (\S+)", hdr).group(1)) == ABS_FLOOR
