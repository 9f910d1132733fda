"""Exact value ranges of the lazy Poseidon2 S-box (powdr_amd/csrc/poseidon2.hpp sbox7 / sbox7_lazy): every intermediate must fit
32 bits and every raw product must stay below 2^64 - (2^32 - 1) p, the domain of the lazy Montgomery reduction."""
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
