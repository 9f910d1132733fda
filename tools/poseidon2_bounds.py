"""Exact value ranges of the lazy forms in powdr_amd/csrc/poseidon2.hpp (sbox7 / sbox7_lazy, internal_layer): every intermediate
must fit 32 bits and every raw product must stay below 2^64 - (2^32 - 1) p, the domain of the lazy Montgomery reduction."""
p, R = 0x78000001, 1 << 32


def out_max(tmax):
    assert tmax + (R - 1) * p < 1 << 64, f"product {tmax / p / p:.3f} p^2 leaves the domain of the reduction"
    return (tmax + (R - 1) * p) >> 32


def sbox(xmax, name):
    x2 = out_max(xmax * xmax)
    x3 = out_max(x2 * xmax)
    x4 = out_max(x2 * x2)
    assert x4 < R
    x4r = max(p - 1, x4 - p)
    lazy = out_max(x3 * x4r)
    print(f"{name}: x < {xmax / p:.4f} p | x2 < {x2 / p:.4f} p | x3 < {x3 / p:.4f} p | x4 < {x4 / p:.4f} p -> < {x4r / p:.4f} p | "
          f"x3*x4 < {x3 * x4r / p / p:.3f} p^2 | lazy x^7 < {lazy / p:.4f} p")
    return lazy


loose = int((1 - 273 / (2 ** 39 / p)) * 128 * p + p + 273 * 128) + 1  # reduce_wide_loose's output bound
o = sbox(loose, "external rounds (input from reduce_wide_loose)")
print(f"external layer output before its reduction: 5 * (7 * {o / p:.3f} p + p) = {5 * (7 * o + p) / p:.1f} p  (reduce_wide takes < 128 p)")
sbox(p - 1, "partial rounds (canonical input)")


# partial rounds: s_0 = lazy S-box output, the other fifteen words lazy reductions of sum * R + mu_i * s_i
R_MOD_P = R % p
s0 = o                      # exclusive bound of s_0 when it enters internal_layer
B = loose                   # the words enter the first partial round from reduce_wide_loose
for it in range(200):
    wide = (s0 - 1) + 15 * (B - 1)
    assert wide < 128 * p, "16-term sum leaves reduce_wide_loose's domain"
    t = (loose - 1) * R_MOD_P + (B - 1) * (p - 1)
    nb = out_max(t) + 1
    assert nb < R
    if nb <= B:
        break
    B = nb
t0 = (loose - 1) * R_MOD_P + (s0 - 1) * (p - 1) + (p - 1) * R_MOD_P  # with the next constant folded in
assert out_max(t0) < 2 * p  # s_0 leaves canonical after ONE conditional subtraction
print(f"partial rounds: words < {B / p:.5f} p (fixed point after {it} rounds), sum < {((s0 - 1) + 15 * (B - 1)) / p:.2f} p, "
      f"product < {((loose - 1) * R_MOD_P + (B - 1) * (p - 1)) / p / p:.4f} p^2 (domain {((1 << 64) - (R - 1) * p) / p / p:.4f} p^2)")
# leaving the partial rounds: reduce_2p(word) + round constant, reduced once, must be an S-box input
exit_max = max(p - 1, B - 1 - p) + (p - 1)
assert exit_max < R
exit_max = max(p - 1, exit_max - p)
assert exit_max < loose, "exit of the partial rounds exceeds the S-box input range"
print(f"exit of the partial rounds: < {(exit_max + 1) / p:.5f} p")
# first partial round: s_0 from reduce_wide_loose + constant, one conditional subtraction (add_loose)
a = (loose - 1) + (p - 1)
assert a < R and max(p - 1, a - p) < loose
