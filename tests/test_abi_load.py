"""The C-ABI library loads and exports every symbol include/*.h declares (no GPU needed)."""
import ctypes
import re
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent


def declared_symbols(header: Path):
    text = header.read_text()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(_apc_\w+|powdr_\w+|pw_\w+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    from powdr_amd import abi

    for header in sorted((ROOT / "include").glob("*.h")):
        syms = declared_symbols(header)
        assert syms, header
        for s in syms:
            assert hasattr(abi.lib, s), f"{s} declared in {header.name} but not exported"
    assert b"gfx950" in abi.lib.powdr_gpu_version()


def test_struct_layouts_match_reference_abi():
    from powdr_amd import abi

    # /root/reference/openvm/src/cuda_abi.rs:66-95,149-169 (#[repr(C)])
    assert ctypes.sizeof(abi.OriginalAir) == 24 and abi.OriginalAir.buffer.offset == 8
    assert abi.OriginalAir.row_block_size.offset == 16
    assert ctypes.sizeof(abi.Subst) == 16
    assert ctypes.sizeof(abi.DerivedExprSpec) == 16 and abi.DerivedExprSpec.span.offset == 8
    assert ctypes.sizeof(abi.DevInteraction) == 12
    assert ctypes.sizeof(abi.ExprSpan) == 8


def test_field_helpers_selftest():
    """Range-sensitive arithmetic helpers of the kernels (lazy / loose reductions, wide accumulators, mul2, the
    extension field) against plain modular arithmetic, including the edges of the documented ranges."""
    from powdr_amd import abi

    abi.lib.powdr_field_selftest.restype = ctypes.c_int
    abi.lib.powdr_field_selftest.argtypes = [ctypes.c_uint64, ctypes.c_uint32]
    for seed in (1, 2, 0xDEADBEEF):
        assert abi.lib.powdr_field_selftest(seed, 20000) == 0


def test_poseidon2_range_analysis_holds():
    """tools/poseidon2_bounds.py recomputes, with exact integer arithmetic, every range the signed Poseidon2 (csrc/poseidon2.hpp)
    relies on — int32, the domain of the signed Montgomery reduction, the wide reductions — and asserts them."""
    import subprocess
    import sys
    from pathlib import Path

    root = Path(__file__).resolve().parents[1]
    out = subprocess.run([sys.executable, str(root / "tools" / "poseidon2_bounds.py")], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr[-1500:]
    assert "partial rounds" in out.stdout and "last layer" in out.stdout
