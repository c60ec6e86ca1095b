"""CPU checks of the exactness arguments the BC7/BC6H kernels rely on (csrc/bc7_exact.hpp, csrc/bc6h.hip).

The kernels replace fp32 arithmetic of the reference by integer / re-derived forms wherever the fp32 arithmetic
provably never rounds.  GPU parity against the oracle covers them end to end; these tests pin the individual claims
with exact rational arithmetic, including the worst cases the proofs identify, so that a broken constant or bound
shows up here (no GPU needed) and not as one wrong block in ten million.
"""
import os
import re
from fractions import Fraction

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "intel-texture-works-plugin_amd", "csrc")


def rn32(x):
    """Round a Fraction to the nearest float32 (ties to even), returned as an exact Fraction."""
    if x == 0:
        return Fraction(0)
    s = -1 if x < 0 else 1
    a = abs(x)
    e = a.numerator.bit_length() - a.denominator.bit_length()
    if Fraction(2) ** e > a:
        e -= 1
    # now 2^e <= a < 2^(e+1); 24-bit significand: quantum 2^(e-23)
    q = Fraction(2) ** (e - 23)
    n = a / q
    f = n.numerator // n.denominator
    r = n - f
    if r > Fraction(1, 2) or (r == Fraction(1, 2) and (f & 1)):
        f += 1
    return s * f * q


def reference_index(N, D, levels):
    """kernel.ispc:1158-1161 in exact arithmetic: p = RN(N/D) (true divide), x = RN(p*levels + 0.5), (int)x, clamp."""
    if D == 0:
        return 1                                  # 0/0 = NaN -> cvttps2dq INT_MIN -> clamp
    p = rn32(Fraction(N, D))
    x = rn32(p * levels + Fraction(1, 2))
    t = int(x)                                    # truncation toward zero
    return min(max(t, 1), levels - 1)


def kernel_index_biased(N, D, levels):
    """select_texel, BITS <= 3: x~ = fma(M, k0, k1), M = -N, k0 = RN(levels / -D), k1 = RN(0.5 + RN(-0.25 / -D))."""
    if D == 0:
        k0, k1 = Fraction(0), Fraction(1, 2)
    else:
        dn = Fraction(-D)
        k0 = rn32(Fraction(levels) / dn)
        k1 = rn32(Fraction(1, 2) + rn32(Fraction(-1, 4) / dn))
    x = rn32(Fraction(-N) * k0 + k1)              # one FMA = one rounding
    t = int(x)
    return min(max(t, 1), levels - 1)


def kernel_index_markstein(N, D, levels):
    """select_texel, BITS == 4: q0 = RN(M*rn); rem = fma(-q0, dn, M); q = fma(rem, rn, q0); x = fma(q, levels, 0.5)."""
    if D == 0:
        dn, r = Fraction(0), Fraction(0)
    else:
        dn = Fraction(-D)
        r = rn32(1 / dn)
    m = Fraction(-N)
    q0 = rn32(m * r)
    rem = rn32(-q0 * dn + m)
    assert D == 0 or rem == -q0 * dn + m          # the remainder is exact in fp32
    q = rn32(rem * r + q0)
    if D:
        assert q == rn32(Fraction(N, D))          # = the true divide
    x = rn32(q * levels + Fraction(1, 2))
    t = int(x)
    return min(max(t, 1), levels - 1)


def _cases(levels, dmax, rng, n_random):
    """(N, D) pairs: the proof's worst cases (y + 0.5 just below / at an integer, largest D) and random ones."""
    out = []
    ds = [1, 2, 3, 5, 7, 255, 256, 65025, dmax, dmax - 1, dmax - 2, 195075, 130050, 3 * 127 * 127]
    ds += [int(v) for v in rng.integers(1, dmax + 1, size=120)]
    for D in ds:
        if D > dmax:
            continue
        for m in range(1, levels + 1):
            base = ((2 * m - 1) * D) // (2 * levels)          # y + 0.5 crosses m at N = (2m-1) D / (2 levels)
            for dN in (-2, -1, 0, 1, 2):
                out.append((base + dN, D))
        out.append((0, D)); out.append((-D, D)); out.append((3 * D, D)); out.append((-260100, D)); out.append((260100, D))
    Ds = rng.integers(1, dmax + 1, size=n_random)
    Ns = rng.integers(-260100, 260101, size=n_random)
    out += list(zip(Ns.tolist(), Ds.tolist()))
    out.append((0, 0))
    return out


@pytest.mark.parametrize("levels,dmax", [(4, 260100), (8, 195075)])
def test_biased_fma_index_equals_true_divide_index(levels, dmax):
    """2-bit indices occur with 3 or 4 channels (D <= 4*255^2), 3-bit indices with 3 channels only (D <= 3*255^2)."""
    rng = np.random.default_rng(levels)
    for N, D in _cases(levels, dmax, rng, 6000):
        assert kernel_index_biased(N, D, levels) == reference_index(N, D, levels), (N, D, levels)


def test_markstein_index_equals_true_divide_index():
    rng = np.random.default_rng(16)
    for N, D in _cases(16, 260100, rng, 4000):
        assert kernel_index_markstein(N, D, 16) == reference_index(N, D, 16), (N, D)


def test_reference_index_is_floor_of_exact_ratio():
    """Claim (1) of the header: the reference's index = clamp(floor(N*levels/D + 1/2)) for every worst case."""
    rng = np.random.default_rng(5)
    for levels, dmax in ((4, 260100), (8, 260100), (16, 260100)):
        for N, D in _cases(levels, dmax, rng, 2000):
            if D == 0:
                continue
            y = Fraction(N * levels, D) + Fraction(1, 2)
            fl = y.numerator // y.denominator
            assert reference_index(N, D, levels) == min(max(fl, 1), levels - 1), (N, D, levels)


def palette_index_2bit(N, D):
    """pal_lower_level<2> of bc7_exact.hpp (round 3): q1 = 1 + [N >= ceil(3D/8)] + [N >= ceil(5D/8)], each [..] the sign bit of
    (th - 1 - N) in 32-bit arithmetic; D == 0 -> thresholds no N reaches."""
    if D == 0:
        th1 = th2 = 0x3fffffff
    else:
        th1, th2 = ((3 * D + 7) >> 3) - 1, ((5 * D + 7) >> 3) - 1
    bit = lambda v: ((v & 0xffffffff) >> 31)
    assert -2**31 <= th1 - N < 2**31 and -2**31 <= th2 - N < 2**31            # no wrap
    return 1 + bit(th1 - N) + bit(th2 - N)


def palette_index_3bit(N, D):
    """pal_lower_level<3>: x' = fma(N, k0, k1'), k0 = -(8 * r), k1' = -0.25 * r, r = RN(1 / -D); the low mantissa bits of
    RN(x' + (1.5 * 2^23 - 1)) are floor(y + 0.5) - 1, clamped to [0, 6] on the raw word."""
    if D == 0:
        k0 = k1 = Fraction(0)
    else:
        r = rn32(1 / Fraction(-D))
        k0, k1 = -(8 * r), -(r / 4)
        assert rn32(k0) == k0 and rn32(k1) == k1                                # exact scalings
    x = rn32(Fraction(N) * k0 + k1)                                             # one FMA
    s = rn32(x + 12582911)
    assert 2**23 <= s < 2**24 and s.denominator == 1                            # ulp 1: the add rounds to an integer
    word = 0x4b000000 + (int(s) - 2**23)                                        # float32 bits of s
    K = 0x4b400000
    assert 0x4b3fffff == 0x4b000000 + (12582911 - 2**23)
    return min(max(word, K), K + 6) - K + 1


def test_round3_palette_index_forms_equal_the_reference_index():
    """The integer thresholds (2-bit indices) and the magic-number floor (3-bit) of the palette path give the reference's
    index on the proofs' worst cases (y + 0.5 at and next to every integer, extreme D) and on random (N, D)."""
    rng = np.random.default_rng(33)
    for N, D in _cases(4, 260100, rng, 20000):
        assert palette_index_2bit(N, D) == reference_index(N, D, 4), (N, D)
    for N, D in _cases(8, 195075, rng, 8000):
        assert palette_index_3bit(N, D) == reference_index(N, D, 8), (N, D)
    # every D a 3-channel segment of small span can have, all N around all thresholds (dense for small D, where 1/(4D) is large)
    for D in range(1, 400):
        for N in range(-3, 3 * D + 3):
            assert palette_index_2bit(N, D) == reference_index(N, D, 4), (N, D)
            assert palette_index_3bit(N, D) == reference_index(N, D, 8), (N, D)


def test_palette_endpoint_bytes_give_the_projection():
    """N = sum (t-a)(b-a) = t.b - t.a - a.(b-a) (pal_project): two unsigned byte dot products and a constant, in 32-bit wrap-around."""
    rng = np.random.default_rng(34)
    a = rng.integers(0, 256, size=(5000, 4)); b = rng.integers(0, 256, size=(5000, 4)); t = rng.integers(0, 256, size=(5000, 4))
    n_ref = ((t - a) * (b - a)).sum(axis=1)
    nc = (-(a * (b - a)).sum(axis=1)) & 0xffffffff
    n = (((t * b).sum(axis=1) + nc) - (t * a).sum(axis=1)) & 0xffffffff
    n = np.where(n >= 2**31, n - 2**32, n)
    assert np.array_equal(n, n_ref)


def _header(name):
    return open(os.path.join(CSRC, name)).read()


def test_weight_tables_in_the_perm_constants():
    """The v_perm byte tables and the 4-bit formula of bc7_exact.hpp hold the format's interpolation weights
    (kernel.ispc:675-686 = BC6HBC7.cpp:35-37)."""
    w2, w3 = [0, 21, 43, 64], [0, 9, 18, 27, 37, 46, 55, 64]
    w4 = [0, 4, 9, 13, 17, 21, 26, 30, 34, 38, 43, 47, 51, 55, 60, 64]
    src = _header("bc7_exact.hpp")
    assert "0x402b1500u" in src and "0x40372e25u, 0x1b120900u" in src
    assert [(0x402b1500 >> (8 * i)) & 255 for i in range(4)] == w2
    assert [((0x40372e25 << 32 | 0x1b120900) >> (8 * i)) & 255 for i in range(8)] == w3
    assert [((q * 68 + 8) >> 4) for q in range(16)] == w4
    for bits, tab in ((2, w2), (3, w3), (4, w4)):             # and the generic formula used by the float paths
        d = (1 << bits) - 1
        assert [(q * 128 + d) // (2 * d) for q in range(1 << bits)] == tab


def test_decode_identity():
    """(int)(((64-w)*a + w*b + 32)/64) = a + ((w*(b-a) + 32) >> 6) for all 8-bit and a sample of 16-bit endpoints."""
    w = np.array([0, 4, 9, 13, 17, 21, 26, 30, 34, 38, 43, 47, 51, 55, 60, 64, 18, 27, 37, 46], dtype=np.int64)
    a = np.arange(256, dtype=np.int64)
    A, B, W = np.meshgrid(a, a, w, indexing="ij")
    assert np.array_equal(((64 - W) * A + W * B + 32) // 64, A + ((W * (B - A) + 32) >> 6))
    rng = np.random.default_rng(3)
    A = rng.integers(0, 65536, size=200000); B = rng.integers(0, 65536, size=200000); W = rng.choice(w, size=200000)
    lhs = ((64 - W) * A + W * B + 32) // 64
    assert np.array_equal(lhs, A + ((W * (B - A) + 32) >> 6))
    # and the fp32 route of the BC6H kernel: every intermediate is an integer below 2^24, floor is exact
    f = np.float32
    rhs = f(A) + np.floor((f(W) * (f(B) - f(A)) + f(32.0)) * f(0.015625))
    assert np.array_equal(lhs.astype(np.float32), rhs.astype(np.float32))
    assert (np.abs(W * (B - A)) + 32 < 2 ** 23).all()


def test_rcp_of_count_table(oracle):
    """RCP_OF_COUNT[n] (scalar constant in table-order scans) = the pinned ISPC rcp(n)."""
    src = _header("bc7_exact.hpp")
    body = src[src.index("RCP_OF_COUNT[17]"):]
    body = body[body.index("{") + 1:body.index("}")]
    vals = [int(v, 16) for v in re.findall(r"0x([0-9a-fA-F]{8})u", body)]
    assert len(vals) == 17
    L = oracle.lib()
    for n in range(1, 17):
        got = np.array([L.oracle_rcp(float(n))], dtype=np.float32).view(np.uint32)[0]
        assert vals[n] == int(got), n


def test_bytemask_table_matches_subset_masks():
    """BCN_BYTEMASK is BCN_SUBSET_MASKS re-expressed over the planar (0,2,4,6)(1,3,5,7)(8,10,12,14)(9,11,13,15) layout."""
    def arr(text, name, n):
        body = text[text.index(name):]
        body = body[body.index("{") + 1:body.index("}")]
        v = [int(x, 16) for x in re.findall(r"0x([0-9a-fA-F]+)u", body)]
        assert len(v) == n, (name, len(v))
        return v
    masks = arr(_header("bc7_tables.h"), "BCN_SUBSET_MASKS[128]", 128)
    bm = arr(_header("bc7_bytemasks.h"), "BCN_BYTEMASK[1024]", 1024)
    eo = [[0, 2, 4, 6], [1, 3, 5, 7], [8, 10, 12, 14], [9, 11, 13, 15]]
    for s, m in enumerate(masks):
        for sub, bits in enumerate((m & 0xffff, m >> 16)):
            for d in range(4):
                want = sum(0xff << (8 * i) for i, k in enumerate(eo[d]) if (bits >> k) & 1)
                assert bm[s * 8 + sub * 4 + d] == want, (s, sub, d)


def test_integer_moment_sums_are_exact_in_fp32():
    """Bounds used by stats_int / refit_line: every partial sum of the reference's float accumulation is an integer
    below 2^24, so the accumulation order cannot matter."""
    assert 16 * 255 * 255 < 2 ** 24                      # second moments of a subset
    assert (16 * 255) ** 2 < 2 ** 24                     # sum_a * sum_b in covariance_from_stats
    assert 16 * 15 * 255 < 2 ** 24                       # sum (L-1-q) * t of opt_endpoints
    assert 16 * 4 * 255 * 255 < 2 ** 24                  # block error incl. alpha
    rng = np.random.default_rng(9)
    t = rng.integers(0, 256, size=(2000, 16, 2)).astype(np.float32)
    acc = np.zeros(2000, dtype=np.float32)
    for k in rng.permutation(16):                        # any order
        acc += t[:, k, 0] * t[:, k, 1]
    assert np.array_equal(acc, (t[:, :, 0].astype(np.int64) * t[:, :, 1].astype(np.int64)).sum(axis=1).astype(np.float32))


def test_f02_schedule_is_a_valid_cache_program():
    """csrc/bc7_f02_schedule.h (tools/gen_bc7_f02_schedule.py): every three-subset shape is visited exactly once, and
    every `load` finds in its slot the result of the SAME texel mask, stored earlier and not overwritten since."""
    def arr(text, name, n):
        body = text[text.index(name):]
        body = body[body.index("{") + 1:body.index("}")]
        v = [int(x, 16) for x in re.findall(r"0x([0-9a-fA-F]+)u", body)]
        assert len(v) == n, (name, len(v))
        return v
    masks = arr(_header("bc7_tables.h"), "BCN_SUBSET_MASKS[128]", 128)
    sched = arr(_header("bc7_f02_schedule.h"), "BC7_F02_SCHEDULE[64]", 64)
    assert sorted(w & 63 for w in sched) == list(range(64))
    slots = {}
    loads = 0
    for w in sched:
        shape = 64 + (w & 63)
        m0, m1 = masks[shape] & 0xffff, masks[shape] >> 16
        sub = [m0, m1, ~(m0 | m1) & 0xffff]
        assert m0 and m1 and sub[2] and (m0 | m1 | sub[2]) == 0xffff and not (m0 & m1)
        for j in range(3):
            act, slot = (w >> (8 + 4 * j)) & 3, (w >> (10 + 4 * j)) & 3
            assert act in (0, 1, 2) and slot in (0, 1, 2, 3)
            if act == 2:
                assert slots.get(slot) == sub[j], (hex(w), j)
                loads += 1
            elif act == 1:
                slots[slot] = sub[j]
    assert loads >= 46                                    # the point of the exercise: 48 of 192 with four slots


def test_bc6h_float_weight_form_is_exact():
    """bc6h.hip select_hdr: the interpolation weights (q*128 + D) / (2*D) of the format (D = 7, 15) computed as
    floor(fma(q, 64/D, 1/2)) and, for the level below, floor(fma(q, 64/D, 1/2 - 64/D)) in fp32 -- equal for every q, with a
    margin far above the rounding error of one fma."""
    f = np.float32
    for bits in (3, 4):
        d = (1 << bits) - 1
        step = f(64.0) / f(d)
        for q in range(1, d + 1):
            want1, want0 = (q * 128 + d) // (2 * d), ((q - 1) * 128 + d) // (2 * d)
            x1 = float(f(q)) * float(step) + 0.5                     # the fma evaluates this exactly before one rounding
            x0 = float(f(q)) * float(step) + float(f(0.5) - step)
            assert int(np.floor(f(x1))) == want1 and int(np.floor(f(x0))) == want0, (bits, q)
            for x in (x1, x0):
                assert min(x - np.floor(x), np.ceil(x) - x) > 0.02 or x == np.floor(x), (bits, q, x)


def test_round4_expand_to_byte_in_the_float_domain():
    """csrc/bc7.hip quant_pbit<0> / quant_shared_pbit (scans): a code c is carried as an integer-valued float and its byte
    expansion expand_to_byte(c, BITS) = (c << (8 - BITS)) + ((c << (8 - BITS)) >> BITS) (kernel.ispc:976-981) is formed as
    c * 8 + floor(c * 0.25) for 5 bits (mode 0: 4 bits + p-bit) and c * 2 + floor(c * (1/64)) for 7 bits (mode 1: 6 bits + shared
    p-bit).  Every code, in fp32 exactly as the kernel evaluates it."""
    import numpy as np
    f = np.float32
    for bits, scale, frac in ((5, f(8.0), f(0.25)), (7, f(2.0), f(0.015625))):
        for c in range(1 << bits):
            vv = c << (8 - bits)
            want = vv + (vv >> bits)
            cf = f(c)
            got = f(f(cf * scale) + np.floor(f(cf * frac)))
            assert float(got) == float(want) and 0 <= want <= 255, (bits, c, got, want)
    # the hypotheses themselves: for t = e/255 * L2 in [0, L2] the SAFE forms give the reference's clamped codes
    for L2, hi0 in ((31, 30), (127, 126)):
        for k in range(0, 4 * L2 + 1):
            t = f(k) / f(4.0)                                   # quarter steps cover both sides of every rounding boundary
            for b in (0, 1):
                u = f(f(f(t - f(b)) * f(0.5)) + f(0.5))
                v = int(np.trunc(u)) * 2 + b                    # (int) of the reference: truncation (u > -1 here)
                ref = min(max(v, b), L2 - 1 + b)
                if b == 0:
                    got = min(float(np.floor(f(f(t * f(0.5)) + f(0.5))) * 2.0), float(hi0))
                else:
                    got = float(np.floor(f(t * f(0.5)))) * 2.0 + 1.0
                assert got == ref, (L2, float(t), b, got, ref)
