"""CPU model of the arithmetic of csrc/gemm_bx.hip (no GPU, no kernel): an fp32 operand as three bf16 pieces and a product
as the six largest piece products.  The kernels themselves are covered by the `-m gpu` tests; this file pins the CLAIMS
their header makes about the number format, with numpy only:
  * the resident operand's split (every piece rounded to nearest even) is exact: x == x1 + x2 + x3;
  * the streaming operand's split (one rounding, two cuts) loses at most 2^-24 |x|;
  * every bf16 x bf16 piece product is exact in fp32;
  * the six-product sum differs from the exact product by less than 2^-23 |x y| (the dropped products are below
    2^-24 |x y|), so a K-term dot product accumulated in fp32 is as close to the exact sum as an fp32 fma chain."""
import numpy as np


def bf16_rne(x):
    """float32 -> the nearest bf16 (ties to even), returned as float32."""
    u = np.asarray(x, dtype=np.float32).view(np.uint32).astype(np.uint64)
    u = (u + 0x7FFF + ((u >> 16) & 1)) & 0xFFFF0000
    return u.astype(np.uint32).view(np.float32)


def bf16_cut(x):
    """float32 -> bf16 by dropping the low 16 bits (v_perm_b32 of the high halves)."""
    return (np.asarray(x, dtype=np.float32).view(np.uint32) & np.uint32(0xFFFF0000)).view(np.float32)


def split_resident(x):
    x = np.asarray(x, dtype=np.float32)
    x1 = bf16_rne(x)
    r = (x - x1).astype(np.float32)
    x2 = bf16_rne(r)
    x3 = bf16_rne((r - x2).astype(np.float32))
    return x1, x2, x3


def split_streaming(x):
    x = np.asarray(x, dtype=np.float32)
    x1 = bf16_rne(x)
    r = (x - x1).astype(np.float32)
    x2 = bf16_cut(r)
    x3 = bf16_cut((r - x2).astype(np.float32))
    return x1, x2, x3


def _samples(n, seed):
    rng = np.random.default_rng(seed)
    mant = rng.standard_normal(n).astype(np.float32)
    expo = rng.integers(-20, 20, n)
    return (mant * np.exp2(expo)).astype(np.float32)


def test_three_bf16_pieces_carry_an_fp32_value():
    x = _samples(200_000, 1)
    for split, bound in ((split_resident, 0.0), (split_streaming, 2.0 ** -24)):
        x1, x2, x3 = split(x)
        for p in (x1, x2, x3):
            assert np.array_equal(p.view(np.uint32) & 0xFFFF, np.zeros_like(p, dtype=np.uint32))     # bf16 values
        err = np.abs(x.astype(np.float64) - (x1.astype(np.float64) + x2 + x3))
        assert np.all(err <= bound * np.abs(x)), float((err / np.abs(x)).max())
        # the pieces shrink by at least 2^-8 each: what bounds the dropped products
        assert np.all(np.abs(x2) <= 2.0 ** -7 * np.abs(x1) + 1e-45) and np.all(np.abs(x3) <= 2.0 ** -7 * np.abs(x2) + 1e-45)


def test_piece_products_are_exact_and_six_of_them_suffice():
    x, y = _samples(100_000, 2), _samples(100_000, 3)
    a, b = split_streaming(x), split_resident(y)
    exact = x.astype(np.float64) * y.astype(np.float64)
    six = np.zeros_like(exact)
    for i, j in ((0, 0), (0, 1), (1, 0), (1, 1), (0, 2), (2, 0)):
        p32 = (a[i] * b[j]).astype(np.float32)                      # what the matrix core forms
        p64 = a[i].astype(np.float64) * b[j].astype(np.float64)
        assert np.array_equal(p32.astype(np.float64), p64)          # 8-bit x 8-bit significands: exact in fp32
        six += p64
    rel = np.abs(six - exact) / np.abs(exact)
    assert float(rel.max()) < 2.0 ** -23, float(rel.max())
    dropped = sum(np.abs(a[i].astype(np.float64) * b[j]) for i, j in ((1, 2), (2, 1), (2, 2)))
    assert float((dropped / np.abs(exact)).max()) < 2.0 ** -22


def test_dot_products_are_as_close_as_an_fp32_chain():
    rng = np.random.default_rng(4)
    K, n = 330, 2000
    X = rng.standard_normal((n, K)).astype(np.float32)
    W = (rng.standard_normal(K) / 18).astype(np.float32)
    exact = X.astype(np.float64) @ W.astype(np.float64)
    a, b = split_streaming(X), split_resident(W)
    main = np.zeros(n, dtype=np.float32)
    corr = np.zeros(n, dtype=np.float32)
    chain = np.zeros(n, dtype=np.float32)
    for k in range(K):                                               # fp32 accumulation, big term and corrections apart
        main = (main + a[0][:, k] * b[0][k]).astype(np.float32)
        for i, j in ((0, 1), (1, 0), (1, 1), (0, 2), (2, 0)):
            corr = (corr + a[i][:, k] * b[j][k]).astype(np.float32)
        chain = np.float32(np.float64(chain) + np.float64(X[:, k]) * np.float64(W[k])).astype(np.float32)   # fma chain
    split = (main + corr).astype(np.float32)
    e_split, e_chain = np.abs(split - exact).mean(), np.abs(chain - exact).mean()
    assert e_split <= 1.1 * e_chain, (e_split, e_chain)
