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
    assert ymax < 128 * p  # floor(y / 128) must fit an int32
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

# external layer: inputs |x| < X, constants centred (|a|, |b|, |d| <= p/2): block outputs < 7 X + 1.5 p, column sums x 5.
# Every layer but the last reduces with a signed Montgomery reduction (|out| < |y| / 2^32 + p / 2); the last one (bias 4 p seeds,
# Barrett-style sreduce_wide_loose) is checked at the end.
def ext_y(X, consts=True):
    return 5 * (7 * X + (3 * HALF if consts else 0))

# closure of "permutation input -> layer -> S-box -> layer ...": inputs are canonical words or sponge outputs ([0, 1.011 p))
lo_last, hi_last = sreduce(35 * p + 60 * p)  # the last layer's outputs (bias included, non-negative): what a sponge hands to the next permutation
sb_in = hi_last
for _ in range(8):
    y0 = ext_y(sb_in)                       # a layer fed by permutation inputs
    x7 = smont(smont(smont(sb_in * sb_in) * sb_in) * smont(smont(sb_in * sb_in) ** 2))
    y1 = ext_y(x7)                          # a layer fed by S-box outputs
    nxt = max(sb_in, smont(y0), smont(y1))
    if nxt == sb_in:
        break
    sb_in = nxt
assert max(y0, y1) < 1 << 62
x7 = sbox(sb_in, "external rounds (inputs: permutation input / Montgomery-reduced layer outputs)")
mid = max(smont(y0), smont(y1))
x7m = sbox(mid, "external rounds after the first layer")
print(f"external layer: |y| < {max(y0, y1) / p:.1f} p, Montgomery-reduced to |x| < {mid / p:.4f} p")
assert sb_in <= lo_ok

# entry of the partial rounds: s_0 + entry_c (centred)
y3 = ext_y(x7m, consts=False)
B0 = smont(y3)
s0_in = B0 + HALF
assert s0_in <= lo_ok
s0 = sbox(s0_in, "partial rounds, s_0 on entry")

# partial rounds: |kappa|, |rho|, |m_i| <= p/2
Bs = [B0]
s0_max, S_max, t_max = s0, 0, 0
for r in range(13):
    B = Bs[-1]
    wide = s0_max * HALF + 15 * B
    S = smont(wide)
    sum_r = S * HALF
    last = r == 12
    t_i = sum_r + (HALF * R_MOD_P if last else 0) + B * HALF
    t_0 = sum_r + HALF * R_MOD_P + s0_max * HALF
    Bs.append(smont(t_i))
    s0_in_next = smont(t_0)
    assert s0_in_next <= lo_ok
    s0_max = max(s0_max, smont(smont(smont(s0_in_next ** 2) * s0_in_next) * smont(smont(s0_in_next ** 2) ** 2)))
    S_max, t_max = max(S_max, S), max(t_max, t_i, t_0)
print(f"partial rounds: |s_i| < {Bs[0] / p:.4f} p on entry, {Bs[1] / p:.4f} p after one round, {max(Bs) / p:.4f} p at most, "
      f"{Bs[13] / p:.4f} p after the last (exit constants added); |sum| < {S_max / p:.4f} p, s_0 leaves below {s0_in_next / p:.4f} p, "
      f"products < {t_max / p / p:.3f} p^2 (domain 1.209 p^2)")
assert max(Bs) <= lo_ok and s0_in_next <= lo_ok, "the words leaving the partial rounds must be S-box inputs of the external rounds"
x7e = sbox(max(Bs[13], s0_in_next), "external round 4 (inputs from the partial rounds)")
assert ext_y(x7e) < 1 << 62

# last layer: bias seeds 4 p -> 40 p or 60 p on an output; the sum itself is within 35 x^7
x7l = x7m
assert 35 * x7l < 40 * p and 35 * x7l + 60 * p < 128 * p
lo_f, hi_f = sreduce(35 * x7l + 60 * p)
print(f"last layer: |sum| < {35 * x7l / p:.1f} p, biased into [{(40 * p - 35 * x7l) / p:.1f} p, {(60 * p + 35 * x7l) / p:.1f} p): outputs in [0, {hi_f / p:.4f} p)")
