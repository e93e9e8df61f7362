"""The split-operand arithmetic of the convolution GEMMs (csrc/conv_gemm.hip split3 / SPLIT,
DESIGN.md section 4.4), modelled in NumPy — no GPU: the statements the default arithmetic rests on.

(1) x == hi + mid + lo EXACTLY for every finite fp32 x whose low part is not in the denormal range,
    with hi, mid, lo bf16 values obtained by round-to-nearest-even and exact fp32 residuals.
(2) The six products kept (hi*hi, hi*mid, mid*hi, mid*mid, hi*lo, lo*hi) are exact in fp32 and the
    three dropped ones (mid*lo, lo*mid, lo*lo) are bounded by 2^-23 |a b| in total — the size of
    one fp32 rounding of the product.
(3) A K-long dot product evaluated that way (16 products summed per MFMA, fp32 accumulator, smallest
    terms first) is at least as close to float64 as a plain fp32 multiply-add loop.
(4) Outside the finite bf16 range the split turns an element into NaN; in the denormal range its
    error is bounded in absolute terms (2^-126 per flushed part)."""
import numpy as np


def bf16_round(x):
    """fp32 -> nearest bf16 (ties to even), returned as fp32."""
    u = np.asarray(x, np.float32).view(np.uint32).astype(np.uint64)
    u = (u + 0x7FFF + ((u >> 16) & 1)) & 0xFFFF0000
    return u.astype(np.uint32).view(np.float32)


def split3(x):
    x = np.asarray(x, np.float32)
    hi = bf16_round(x)
    r1 = (x - hi).astype(np.float32)
    mid = bf16_round(r1)
    r2 = (r1 - mid).astype(np.float32)
    lo = bf16_round(r2)
    return hi, mid, lo


def test_three_bf16_values_carry_every_significand_bit():
    rng = np.random.RandomState(0)
    x = np.concatenate([
        rng.standard_normal(200000).astype(np.float32),
        (rng.standard_normal(50000) * 1e-6).astype(np.float32),
        (rng.standard_normal(50000) * 1e6).astype(np.float32),
        np.float32([0., -0., 1., -1., 255.5, 3.38e38, 1.17549435e-38 * 2 ** 20]),   # (bf16 max 3.3895e38: larger
        # finite fp32 values round to inf in the first conversion, as in the kernel)
        np.nextafter(np.float32(1), np.float32(2)) * rng.uniform(1, 2, 1000).astype(np.float32),
    ])
    hi, mid, lo = split3(x)
    # residuals are exact: |x - hi| <= half an ulp of bf16(x), representable in fp32
    assert np.array_equal((hi.astype(np.float64) + mid + lo).astype(np.float32), x)
    assert np.array_equal(hi.astype(np.float64) + mid.astype(np.float64) + lo.astype(np.float64),
                          x.astype(np.float64))
    # each part has at most 8 significant bits (it IS a bf16 value)
    for part in (hi, mid, lo):
        assert np.all((part.view(np.uint32) & 0xFFFF) == 0)
    # magnitudes: mid <= 2^-8 |x|, lo <= 2^-16 |x|
    nz = x != 0
    assert np.all(np.abs(mid[nz]) <= np.abs(x[nz]) * 2.0 ** -8)
    assert np.all(np.abs(lo[nz]) <= np.abs(x[nz]) * 2.0 ** -16)


def test_dropped_cross_terms_are_one_fp32_rounding():
    rng = np.random.RandomState(1)
    a = rng.standard_normal(100000).astype(np.float32)
    b = rng.standard_normal(100000).astype(np.float32)
    ah, am, al = (t.astype(np.float64) for t in split3(a))
    bh, bm, bl = (t.astype(np.float64) for t in split3(b))
    kept = ah * bh + ah * bm + am * bh + am * bm + ah * bl + al * bh
    exact = a.astype(np.float64) * b.astype(np.float64)
    assert np.all(np.abs(kept - exact) <= np.abs(exact) * 2.0 ** -23)
    # and every kept product is exact in fp32 (8 x 8 significant bits)
    for p, q in ((ah, bh), (ah, bm), (am, bh), (am, bm), (ah, bl), (al, bh)):
        assert np.array_equal((p * q).astype(np.float32).astype(np.float64), p * q)


def _dot_split(a, b):
    """The SPLIT kernel's evaluation order for one output element: per 16-deep K step six MFMAs
    (smallest terms first), each adding the exact sum of its 16 products to the fp32 accumulator."""
    ah, am, al = (t.astype(np.float64) for t in split3(a))
    bh, bm, bl = (t.astype(np.float64) for t in split3(b))
    acc = np.float32(0)
    for k in range(0, len(a), 16):
        s = slice(k, k + 16)
        for p, q in ((ah, bl), (al, bh), (am, bm), (am, bh), (ah, bm), (ah, bh)):
            acc = np.float32(np.float64(acc) + np.dot(p[s], q[s]))
    return acc


def _dot_fp32(a, b):
    acc = np.float32(0)
    for x, y in zip(a, b):
        acc = np.float32(np.float64(acc) + np.float64(np.float32(x) * np.float32(y)))   # fma-free fp32 loop
    return acc


def test_split_dot_product_is_at_least_as_close_to_float64_as_an_fp32_loop():
    rng = np.random.RandomState(2)
    err_s, err_f = [], []
    for trial in range(200):
        K = 512
        a = np.maximum(rng.standard_normal(K), 0).astype(np.float32)      # post-ReLU activations
        b = (rng.standard_normal(K) / np.sqrt(K)).astype(np.float32)
        ref = np.dot(a.astype(np.float64), b.astype(np.float64))
        scale = np.sqrt(np.sum((a.astype(np.float64) * b) ** 2))          # size of the summands
        err_s.append(abs(float(_dot_split(a, b)) - ref) / scale)
        err_f.append(abs(float(_dot_fp32(a, b)) - ref) / scale)
    rms_s, rms_f = np.sqrt(np.mean(np.square(err_s))), np.sqrt(np.mean(np.square(err_f)))
    assert rms_s <= rms_f, (rms_s, rms_f)
    # (both are dominated by the roundings of the fp32 ACCUMULATOR; the worst cases are alike)
    assert max(err_s) <= 1.5 * max(err_f) and max(err_s) <= 3e-6      # far inside the north star's 1e-4


def test_non_finite_and_overflow_semantics_of_the_split():
    """What the kernel's split3 does outside the finite bf16 range (documented next to
    functions.conv.set_gemm_arithmetic, asserted on the device by tests/test_gpu_split_bf16.py):
    an infinite element, and a finite one that rounds to infinity in bf16 (|x| above bf16's largest
    value 3.3895e38 by more than half a bf16 ulp), has hi = +-inf and a NaN among mid / lo: every
    output that reads the element is NaN where fp32 arithmetic would give +-inf.  Everything up to
    0.99 FLT_MAX splits exactly."""
    with np.errstate(invalid='ignore', over='ignore'):
        for bad in (np.inf, -np.inf, np.nan, np.float32(3.4e38), -np.finfo(np.float32).max):
            hi, mid, lo = split3(np.float32([bad]))
            # (a finite overflow has mid = x - inf = -+inf and lo = NaN; the products hi*b and mid*b
            # then cancel to NaN in the accumulator just the same)
            assert not np.isfinite(hi[0]) and (np.isnan(mid[0]) or np.isnan(lo[0]))
    x = np.float32([0.9, -0.99, 0.5]) * np.finfo(np.float32).max
    hi, mid, lo = split3(x)
    assert np.all(np.isfinite(hi)) and np.array_equal(hi.astype(np.float64) + mid + lo, x.astype(np.float64))


def test_denormal_range_error_is_absolute_not_relative():
    """Parts of the split below the smallest normal fp32 / bf16 value 2^-126 may be flushed by the
    bf16 conversion and the matrix pipe: the representation error of an element is then < 2^-126 in
    ABSOLUTE terms whatever is flushed (relative to a tiny element it can be large: below 3e-36 the
    mid part is denormal too).  Modelled here as flush-to-zero of every denormal part."""
    rng = np.random.RandomState(5)
    x = (rng.standard_normal(100000) * 10.0 ** rng.uniform(-44, -30, 100000)).astype(np.float32)
    tiny = np.float32(2.0 ** -126)
    parts = [np.where(np.abs(p) < tiny, np.float32(0), p) for p in split3(x)]
    back = parts[0].astype(np.float64) + parts[1] + parts[2]
    assert np.all(np.abs(back - x.astype(np.float64)) < 3 * 2.0 ** -126)
