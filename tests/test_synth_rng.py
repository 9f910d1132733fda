"""powdr_amd/synth_csrc/synth_rng.c restates numpy's `Generator(PCG64).integers(0, bound, size, dtype=uint32)` in C (the synthetic
inputs' generator is the slowest part of many tests and of the bench's CPU baseline set-up): bit for bit the same draws, the same
generator state afterwards — so no golden digest that depends on these inputs moves."""
import numpy as np
import pytest

from powdr_amd import synth

P = 0x78000001


@pytest.mark.parametrize("bound", [2, 3, 256, 2048, 1 << 17, P, 0xFFFFFFFF, (1 << 31) + 12345])
def test_bounded_draws_equal_numpy(bound):
    for seed, n, pre in ((0, 1 << 16, 0), (1, 100003, 1), (977, (1 << 24) + 5, 3), (5, 65537, 2)):
        a, b = np.random.Generator(np.random.PCG64(seed)), np.random.Generator(np.random.PCG64(seed))
        for g in (a, b):  # an odd number of 32-bit draws first: the buffered half of a 64-bit output is part of the state
            if pre:
                g.integers(0, 1000, size=pre, dtype=np.uint32)
        want = a.integers(0, bound, size=n, dtype=np.uint32)
        got = synth._bounded_u32(b, bound, n)
        assert synth._RNG_LIB, "libpowdr_synth_rng.so was not loaded (python -m powdr_amd.build)"
        assert got.dtype == np.uint32 and (got == want).all()
        # ... and whatever numpy draws next continues the same stream
        assert (a.integers(0, P, size=33, dtype=np.uint32) == b.integers(0, P, size=33, dtype=np.uint32)).all()
        assert a.random() == b.random()


def test_dummy_traces_are_what_numpy_alone_would_draw(monkeypatch):
    s = synth.generate("T1", seed=3)
    fast, dims = synth.fill_dummy_traces_numpy(s, 70000, seed=5)
    monkeypatch.setattr(synth, "_RNG_LIB", False)
    slow, dims2 = synth.fill_dummy_traces_numpy(s, 70000, seed=5)
    assert dims == dims2 and all((x == y).all() for x, y in zip(fast, slow))
