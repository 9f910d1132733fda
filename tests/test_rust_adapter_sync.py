"""rust/powdr-hip (the Rust side of boundaries B1/B2) cannot be compiled in this image (no cargo/rustc), so this test
keeps it honest against the C headers it binds: every symbol a header declares has an `extern "C"` declaration with the
same number of parameters, nothing extra is declared, and every `#[repr(C)]` struct lists the header struct's fields in
the same order. The first block must also match the reference's own FFI (openvm/src/cuda_abi.rs:8-64) name for name."""
import re
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
FFI = (ROOT / "rust" / "powdr-hip" / "src" / "ffi.rs").read_text()


def c_functions():
    out = {}
    for h in sorted((ROOT / "include").glob("*.h")):
        text = re.sub(r"/\*.*?\*/", "", h.read_text(), flags=re.S)
        for m in re.finditer(r"\b(_apc_\w+|powdr_\w+|pw_\w+)\s*\(([^;{]*?)\)\s*;", text, flags=re.S):
            params = m.group(2).strip()
            n = 0 if params in ("", "void") else params.count(",") + 1
            out[m.group(1)] = n
    return out


def rust_functions():
    out = {}
    for m in re.finditer(r"pub fn (\w+)\s*\(([^;]*?)\)\s*(?:->[^;]+)?;", FFI, flags=re.S):
        params = m.group(2).strip().rstrip(",")
        out[m.group(1)] = 0 if not params else params.count(":")
    return out


def test_every_header_symbol_is_bound_with_the_same_arity():
    c, r = c_functions(), rust_functions()
    hip_runtime = {n for n in r if n.startswith("hip")}
    assert set(c) == set(r) - hip_runtime, (sorted(set(c) - set(r)), sorted(set(r) - hip_runtime - set(c)))
    for name, n in c.items():
        assert r[name] == n, f"{name}: header has {n} parameters, ffi.rs {r[name]}"


def c_struct_fields(name):
    for h in sorted((ROOT / "include").glob("*.h")):
        text = re.sub(r"/\*.*?\*/", "", h.read_text(), flags=re.S)
        m = re.search(r"typedef struct(?:\s+\w+)?\s*\{([^}]*)\}\s*" + name + r"\s*;", text, flags=re.S)
        if m:
            fields = []
            for decl in m.group(1).split(";"):
                decl = decl.strip()
                if not decl:
                    continue
                # "uint32_t a, b" / "const uint32_t* p" / "PowdrAirStats before, after"
                names = [re.sub(r"[\*\s]", "", x).split("[")[0] for x in re.split(r",", decl)]
                names[0] = re.findall(r"(\w+)\s*(?:\[\d+\])?$", decl.split(",")[0].strip())[0]
                fields += names
            return fields
    return None


def rust_struct_fields(name):
    m = re.search(r"pub struct " + name + r"\s*\{([^}]*)\}", FFI, flags=re.S)
    return re.findall(r"pub (\w+)\s*:", m.group(1)) if m else None


def test_repr_c_structs_list_the_header_fields_in_order():
    for name in ("OriginalAir", "Subst", "ExprSpan", "DerivedExprSpec", "DevInteraction", "PowdrDeviceMatrix", "PowdrPeriphery", "PowdrCallMajorAir", "PowdrSubstCM", "PowdrOrigInstr", "PowdrRecordSubst",
                 "PowdrAirStats", "PowdrApcCandidateInfo", "PwStarkConfig", "PwSegmentAir", "PwAirDescription"):
        c, r = c_struct_fields(name), rust_struct_fields(name)
        assert c and r, name
        assert c == r, (name, c, r)
        assert re.search(r"#\[repr\(C\)\]\s*(?:#\[derive\([^)]*\)\]\s*)?pub struct " + name + r"\b", FFI), name


def test_reference_ffi_names_are_unchanged(reference_dir):
    ref = (reference_dir / "openvm" / "src" / "cuda_abi.rs").read_text()
    block = ref[ref.index('extern "C"'):ref.index("#[repr(C)]")]
    ref_fns = {m.group(1): m.group(2) for m in re.finditer(r"pub fn (\w+)\s*\((.*?)\)\s*->", block, flags=re.S)}
    assert set(ref_fns) == {"_apc_tracegen", "_apc_apply_derived_expr", "_apc_apply_bus"}
    mine = rust_functions()
    strip = lambda s: re.sub(r"//[^\n]*", "", s)
    for name, params in ref_fns.items():
        ref_names = re.findall(r"(\w+)\s*:", strip(params))
        my_params = re.search(r"pub fn " + name + r"\s*\((.*?)\)\s*->", FFI, flags=re.S).group(1)
        assert re.findall(r"(\w+)\s*:", my_params) == ref_names, name
        assert mine[name] == len(ref_names)
