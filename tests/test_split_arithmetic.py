"""CPU: the arithmetic identity csrc/pointwise_split.hip rests on, restated in numpy - an fp32 number is EXACTLY the sum of three
bf16 numbers obtained by two round-to-nearest-even conversions and two exact subtractions, and the product of two bf16 numbers is
exact in fp32.  (The kernel's own results are held against float64 on the GPU: tests/test_gpu_f_ops.py::test_pointwise_split_*.)"""
import numpy as np


def bf16_rn(x: np.ndarray) -> np.ndarray:
    """fp32 -> nearest bf16 (ties to even), returned as fp32: what v_cvt_pk_bf16_f32 does for finite inputs"""
    b = np.ascontiguousarray(x, dtype=np.float32).view(np.uint32).astype(np.uint64)
    r = ((b + 0x7FFF + ((b >> 16) & 1)) & 0xFFFF0000).astype(np.uint32)
    return r.view(np.float32)


def split3(v: np.ndarray):
    h = bf16_rn(v)
    r1 = (v - h).astype(np.float32)
    m = bf16_rn(r1)
    l = (r1 - m).astype(np.float32)
    return h, r1, m, l


def _operands(seed: int, n: int) -> np.ndarray:
    g = np.random.default_rng(seed)
    mant = g.integers(0, 1 << 23, size=n, dtype=np.uint32)
    # |v| in [2^-97, 2^123): below ~2^-102 the low pieces fall under 2^-126 and are bf16 denormals (not representable at full
    # width, and the matrix pipe may flush them: an absolute error below 1e-37 |w| per product, see the kernel's header)
    exp = g.integers(30, 250, size=n, dtype=np.uint32)
    sign = g.integers(0, 2, size=n, dtype=np.uint32)
    v = ((sign << 31) | (exp << 23) | mant).view(np.float32)
    edge_m = np.array([0, 1, 0x7FFFFF, 0x7FFFFE, 0x008000, 0x007FFF, 0x018000, 0x00FFFF, 0x7F8000, 0x7F7FFF, 0x000080, 0x00017F, 0x7FFF80],
                      dtype=np.uint32)                                  # ties, all-ones, carries into the next binade
    edges = ((np.uint32(127) << 23) | edge_m).view(np.float32)
    return np.concatenate([v, edges, -edges, edges * np.float32(2.0 ** -60), edges * np.float32(2.0 ** 60), np.zeros(2, np.float32)])


def test_three_bf16_pieces_sum_to_the_fp32_number_exactly():
    v = _operands(0, 400000)
    h, r1, m, l = split3(v)
    d = np.float64
    assert np.array_equal(v.astype(d) - h.astype(d), r1.astype(d)), "v - bf16(v) is not exact in fp32"
    assert np.array_equal(r1.astype(d) - m.astype(d), l.astype(d)), "(v - h) - bf16(v - h) is not exact in fp32"
    assert np.array_equal(bf16_rn(l), l), "the last piece is not a bf16 number"
    assert np.array_equal(h.astype(d) + m.astype(d) + l.astype(d), v.astype(d))
    a = np.abs(v.astype(d))
    assert np.all(np.abs(m.astype(d)) <= 2.0 ** -8 * a) and np.all(np.abs(l.astype(d)) <= 2.0 ** -16 * a)
    for piece in (h, m, l):                                             # 8 significant bits each: the low 16 bits of the pattern are 0
        assert not np.any(piece.view(np.uint32) & 0xFFFF)


def test_piece_products_are_exact_in_fp32_and_nine_of_them_are_the_product():
    x, w = _operands(1, 200000), _operands(2, 200000)[::-1].copy()
    x, w = x * np.float32(2.0 ** -40), w * np.float32(2.0 ** -40)       # keep x * w inside the fp32 range
    # operands in (2^-40, 2^40): even the smallest piece product (l x l', >= 2^-32 of x w) stays a normal fp32 number
    keep = (np.abs(x) > 2.0 ** -40) & (np.abs(w) > 2.0 ** -40) & (np.abs(x) < 2.0 ** 40) & (np.abs(w) < 2.0 ** 40)
    x, w = x[keep], w[keep]
    xs, ws = split3(x), split3(w)
    px, pw = (xs[0], xs[2], xs[3]), (ws[0], ws[2], ws[3])
    d = np.float64
    total = np.zeros(x.shape, d)
    six = np.zeros(x.shape, d)
    for qi, a in enumerate(px):
        for ri, b in enumerate(pw):
            p32 = (a * b).astype(np.float32)                            # what one lane-product of the bf16 MFMA contributes
            assert np.array_equal(p32.astype(d), a.astype(d) * b.astype(d)), "a bf16 x bf16 product is not exact in fp32"
            total += p32.astype(d)
            if qi + ri <= 2:
                six += p32.astype(d)
    exact = x.astype(d) * w.astype(d)                                   # 48 significant bits: exact in float64
    assert np.array_equal(total, exact), "the nine piece products do not add up to the product"
    # the six-product form (GLASS_PW_SPLIT=6, opt-in): what it leaves out is bounded by 2^-23 of the product (an IEEE fp32
    # multiply alone rounds by up to 2^-24) - a bound, not exactness, which is why it is not the default
    nz = exact != 0
    rel = np.abs(six - exact)[nz] / np.abs(exact)[nz]
    assert rel.max() <= 2.0 ** -23 * (1 + 2.0 ** -8), rel.max()
