"""Exact value ranges of the signed forms in powdr_amd/csrc/poseidon2.hpp (sbox7, external_layer, internal_layer, permute).
Every 32-bit intermediate must fit an int32, every argument of the signed Montgomery reduction must satisfy
|t| + 2^31 p < 2^63, every wide sum must stay inside sreduce_wide_loose's 64 p.  All bounds are magnitudes (exclusive)."""
p, R = 0x78000001, 1 << 32
R_MOD_P = R % p
I32 = 1 << 31
HALF = (p - 1) // 2  # largest centred representative


def smont(tmax):
    """|t| <= tmax -> bound of |(t + m p) >> 32| for a signed m in [-2^31, 2^31)"""
    assert tmax + I32 * p < 1 << 63, f"|t| = {tmax / p / p:.3f} p^2 leaves the domain of the signed reduction"
    out = (tmax + I32 * p) // R + 1
    assert out < I32
    return out


def sreduce(ymax):
    """|y| <= ymax -> (lowest, highest) value of y - q p, q = floor(floor(y / 128) * 273 / 2^32); exact extremes are hard,
    so use the analytic envelope: y (1 - 273 p / 2^39) + [0, p + 273 p / 2^32 + 1)"""
    assert ymax < 64 * p
    eps = 1 - 273 * p / 2 ** 39
    lo = -int(ymax * eps) - 1
    hi = int(ymax * eps) + p + (273 * p >> 32) + 2
    assert -I32 < lo and hi < I32
    return lo, hi


def sbox(xmax, name):
    x2 = smont(xmax * xmax)
    x3 = smont(x2 * xmax)
    x4 = smont(x2 * x2)
    x7 = smont(x3 * x4)
    print(f"{name}: |x| < {xmax / p:.4f} p | x2 < {x2 / p:.4f} p | x3 < {x3 / p:.4f} p | x4 < {x4 / p:.4f} p | x^7 < {x7 / p:.4f} p")
    return x7


# the widest S-box input the chain tolerates (int32 + reduction domain), for the record
lo_ok = p
for cand in range(p, int(1.2 * p), p // 1000):
    try:
        x2 = smont(cand * cand); x3 = smont(x2 * cand); x4 = smont(x2 * x2); smont(x3 * x4)
        lo_ok = cand
    except AssertionError:
        break
print(f"S-box input may be as large as {lo_ok / p:.3f} p")

# external layer: inputs |x| < X, constants centred (|a|, |b|, |d| <= p/2): block outputs < 7 X + 1.5 p, column sums x 5
def ext_out(X, consts=True):
    y = 5 * (7 * X + (3 * HALF if consts else 0))
    return sreduce(y), y

# closure of "permutation input -> layer -> S-box -> layer ...": inputs are canonical words or sponge outputs
sb_in = p
for _ in range(8):
    (lo, hi), y0 = ext_out(sb_in)                 # a layer fed by permutation inputs / previous outputs
    x7 = smont(smont(smont(sb_in * sb_in) * sb_in) * smont(smont(sb_in * sb_in) ** 2))
    (lo2, hi2), y1 = ext_out(x7)                  # a layer fed by S-box outputs
    nxt = max(sb_in, -lo, hi, -lo2, hi2)
    if nxt == sb_in:
        break
    sb_in = nxt
x7 = sbox(sb_in, "external rounds")
print(f"external layer: |y| < {max(y0, y1) / p:.1f} p (sreduce_wide_loose takes 64 p), outputs in ({min(lo, lo2) / p:.4f} p, {max(hi, hi2) / p:.4f} p)")
assert sb_in <= lo_ok

# entry of the partial rounds: s_0 + int_rc[0] through min(x, x - p) as unsigned
(lo3, hi3), _ = ext_out(x7, consts=False)
assert hi3 + p - 1 < R  # the unsigned sum does not wrap
s0_in = max(hi3, p - lo3)  # [0, hi3) or (-p + lo3, -p)
assert s0_in <= lo_ok
s0 = sbox(max(s0_in, sb_in), "partial rounds, s_0")

# partial rounds: words |s_i| < B_r in round r (B_0 = what the layer before delivers), s_0 an S-box output
Bs = [max(-lo3, hi3)]
s0_max, S_max, wide_max, t_max = s0, 0, 0, 0
for r in range(13):
    B = Bs[-1]
    wide = s0_max + 15 * B
    slo, shi = sreduce(wide)
    S = max(-slo, shi)
    sum_r = S * R_MOD_P
    last = r == 12
    t_i = sum_r + (HALF * R_MOD_P if last else 0) + B * HALF
    t_0 = sum_r + HALF * R_MOD_P + s0_max * HALF
    Bs.append(smont(t_i))
    s0_in_next = smont(t_0)
    assert s0_in_next <= lo_ok
    s0_max = max(s0_max, smont(smont(smont(s0_in_next ** 2) * s0_in_next) * smont(smont(s0_in_next ** 2) ** 2)))
    S_max, wide_max, t_max = max(S_max, S), max(wide_max, wide), max(t_max, t_i, t_0)
print(f"partial rounds: |s_i| < {Bs[0] / p:.4f} p on entry, {Bs[1] / p:.4f} p after one round, {Bs[12] / p:.4f} p after twelve, "
      f"{Bs[13] / p:.4f} p after the last (exit constants added); |sum| < {S_max / p:.4f} p, 16-term sum < {wide_max / p:.2f} p, "
      f"s_0 leaves below {s0_in_next / p:.4f} p, products < {t_max / p / p:.3f} p^2 (domain 1.209 p^2)")
assert Bs[13] <= sb_in and s0_in_next <= sb_in, "the words leaving the partial rounds must be S-box inputs of the external rounds"

# last layer: bias seeds 4 p -> 40 p or 60 p on an output; the sum itself is within 35 x^7
assert 35 * x7 < 40 * p and 35 * x7 + 60 * p < 128 * p
print(f"last layer: |sum| < {35 * x7 / p:.1f} p, biased into [{(40 * p - 35 * x7) / p:.1f} p, {(60 * p + 35 * x7) / p:.1f} p)")
