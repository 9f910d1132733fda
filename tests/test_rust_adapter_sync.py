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


def _rust_sources():
    return {p.name: p.read_text() for p in sorted((ROOT / "rust" / "powdr-hip" / "src").glob("*.rs"))}


def test_every_ffi_name_the_crate_uses_is_declared():
    """`ffi::name` anywhere in the crate -> a function, struct or constant of ffi.rs (whose functions the first test ties to include/*.h)."""
    declared = set(re.findall(r"pub (?:fn|struct|const|type|static) (\w+)", FFI))
    for name, text in _rust_sources().items():
        code = re.sub(r"//[^\n]*", "", text)
        for used in set(re.findall(r"(?<!core::)(?<!std::)\bffi::(\w+)", code)):
            assert used in declared, f"{name} uses ffi::{used}, which ffi.rs does not declare"


def test_every_isa_member_the_crate_uses_exists(reference_dir):
    """round 3 called `ISA::create_dummy_chip_complex_hip`, which `OpenVmISA` does not have. Every member taken from
    `OpenVmISA` must exist in the reference's isa.rs; every member taken from the crate's own extension trait `OpenVmIsaHip` must be
    declared in src/isa_hip.rs; a bare `ISA::member` is not allowed (it would not say which trait it means)."""
    isa_rs = (reference_dir / "openvm" / "src" / "isa.rs").read_text()
    trait = isa_rs[isa_rs.index("pub trait OpenVmISA"):]
    upstream = set(re.findall(r"\bfn (\w+)", trait)) | set(re.findall(r"\btype (\w+)", trait))
    src = _rust_sources()
    ext = src["isa_hip.rs"]
    ext_trait = ext[ext.index("pub trait OpenVmIsaHip"):]
    own = set(re.findall(r"\bfn (\w+)", ext_trait)) | set(re.findall(r"\btype (\w+)", ext_trait))
    assert own == {"HipBuilder", "create_dummy_chip_complex_hip"} and not (own & upstream)
    n_up = n_own = 0
    for name, text in src.items():
        code = re.sub(r"//[^\n]*", "", text)
        assert not re.search(r"\bISA::\w+\s*\(", code), f"{name}: bare ISA::method(...) — say which trait"
        for m in re.findall(r"<ISA as OpenVmISA>::(\w+)", code):
            assert m in upstream, f"{name}: OpenVmISA has no `{m}` (openvm/src/isa.rs)"
            n_up += 1
        for m in re.findall(r"<ISA as OpenVmIsaHip>::(\w+)", code):
            assert m in own, f"{name}: OpenVmIsaHip has no `{m}`"
            n_own += 1
        for m in re.findall(r"\bISA::(\w+)", code):  # associated types written `ISA::Config` etc.
            assert m in upstream or m in own, f"{name}: ISA::{m} exists in neither trait"
    assert n_up >= 1 and n_own >= 2


def test_record_bridge_matches_the_record_layout_tables():
    """records_from_arena.rs (all thirteen chips, no `Unsupported` arm) against the three other statements of the record
    layout: include/powdr_gpu.h (POWDR_ORIG_* numbers), powdr_amd/original_chips.py (RECORD_WORDS / N_PREV_TS, what the library
    consumes) and oracle/original_chips.py (what the oracle consumes). Per chip kind: the kind constant, the (data words, previous
    timestamps) of `shape_of`, a `walk!` arm with an (adapter, core) record pair, and a `RecordView` impl for exactly that pair whose
    `n_data` / `n_prev` and filled array slots equal the tables'."""
    from oracle import original_chips as ooc
    from powdr_amd import original_chips as oc

    text = _rust_sources()["records_from_arena.rs"]
    code = re.sub(r"//[^\n]*", "", text)
    assert "Unsupported" not in code
    hdr = re.sub(r"/\*.*?\*/", "", (ROOT / "include" / "powdr_gpu.h").read_text(), flags=re.S)
    hdr_kinds = {m.group(1): int(m.group(2)) for m in re.finditer(r"POWDR_ORIG_(\w+) = (\d+)", hdr)}
    assert hdr_kinds.pop("KIND_COUNT") == 13 == oc.N_KINDS and len(hdr_kinds) == 13
    rust_kinds = {m.group(1): int(m.group(2)) for m in re.finditer(r"pub const (\w+): u32 = (\d+);", code)}
    assert rust_kinds == hdr_kinds
    assert list(ooc.RECORD_WORDS) == list(oc.RECORD_WORDS) and [ooc.N_PREV_TS[k] for k in range(13)] == list(oc.N_PREV_TS)
    # shape_of
    shape_fn = code[code.index("fn shape_of"):code.index("pub fn records_from_arenas")]
    shape = {}
    for arm in re.finditer(r"((?:\w+\s*\|\s*)*\w+)\s*=>\s*Ok\(\((\d+), (\d+)\)\)", shape_fn):
        for name in re.split(r"\s*\|\s*", arm.group(1)):
            shape[rust_kinds[name]] = (int(arm.group(2)), int(arm.group(3)))
    want = {k: (oc.RECORD_WORDS[k] - oc.N_PREV_TS[k], oc.N_PREV_TS[k]) for k in range(13)}
    assert shape == want
    # walk! arms: kind -> (adapter, core)
    walk = {rust_kinds[m.group(1)]: (m.group(2), re.sub(r"\s", "", m.group(3))) for m in re.finditer(r"(\w+) => walk!\((\w+), ([\w<>, ]+)\)", code)}
    assert sorted(walk) == list(range(13))
    # RecordView impls (the mult adapter's three come from one macro)
    impls = {}
    for m in re.finditer(r"impl RecordView for \(&(\w+), &([\w<>, $:]+)\) \{(.*?)\n\s*\}\n\s*\}", code, flags=re.S):
        impls[(m.group(1), re.sub(r"\s", "", m.group(2)))] = m.group(3)
    for core in re.findall(r"mult_view!\(([\w<>, ]+)\);", code):
        impls[("Rv32MultAdapterRecord", re.sub(r"\s", "", core))] = impls[("Rv32MultAdapterRecord", "$core")]
    for k in range(13):
        body = impls[walk[k]]
        n_data, n_prev = int(re.search(r"n_data: (\d)", body).group(1)), int(re.search(r"n_prev: (\d)", body).group(1))
        assert (n_data, n_prev) == want[k], (oc.KIND_NAMES[k], walk[k])
        data = [x.strip() for x in re.search(r"data: \[(.*?)\],\s*n_data", body, flags=re.S).group(1).split(",")]
        prev = [x.strip() for x in re.search(r"prev_ts: \[(.*?)\],\s*n_prev", body, flags=re.S).group(1).split(",")]
        # top-level commas only (word(c.prev_data.map(|x| x as u8)) has none): three slots, the unused ones literally 0
        assert len(data) == 3 and len(prev) == 3, (oc.KIND_NAMES[k], data, prev)
        assert [d != "0" for d in data] == [i < n_data for i in range(3)] and [t != "0" for t in prev] == [i < n_prev for i in range(3)]
        assert all("prev_timestamp" in t for t in prev[:n_prev]) and "from_timestamp" in body
    # the adapter each chip uses: the shapes INTEGRATION.md §3b tabulates
    by_adapter = {}
    for k, (ad, _) in walk.items():
        by_adapter.setdefault(ad, set()).add(oc.KIND_NAMES[k])
    assert by_adapter == {"Rv32BaseAluAdapterRecord": {"BaseAlu", "Shift", "LessThan"}, "Rv32MultAdapterRecord": {"Multiplication", "MulH", "DivRem"},
                          "Rv32LoadStoreAdapterRecord": {"LoadStore", "LoadSignExtend"}, "Rv32BranchAdapterRecord": {"BranchEqual", "BranchLessThan"},
                          "Rv32CondRdWriteAdapterRecord": {"JalLui"}, "Rv32RdWriteAdapterRecord": {"Auipc"}, "Rv32JalrAdapterRecord": {"Jalr"}}
    first = [lo for lo, hi, k in sorted(oc.OPCODE_RANGES, key=lambda t: t[2])]
    chip = _rust_sources()["chip.rs"]
    assert "[" + ", ".join(str(x) for x in first) + "]" in chip  # FIRST_OPCODE of air_name_of_kind
