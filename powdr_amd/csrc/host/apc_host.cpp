// Host side of GPU APC trace generation (see include/powdr_host.h for what it mirrors).
#include "../../../include/powdr_host.h"
#include "../common.hpp"
#include "../original_chips_tables.hpp"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <memory>
#include <stdexcept>
#include <string>
#include <unordered_map>
#include <vector>

namespace {

constexpr uint32_t kP = 0x78000001u;

// ---------------------------------------------------------------------------------- JSON
struct JVal;
using JPtr = std::unique_ptr<JVal>;
struct JVal {
    enum Kind { Null, Bool, Num, Str, Arr, Obj } kind = Null;
    bool b = false;
    bool neg = false;
    uint64_t u = 0;  // magnitude of an integer number
    bool is_float = false;
    double d = 0.0;  // value of a non-integer number (cost_before / cost_after of apc_candidates.json; CBOR floats)
    std::string s;
    std::vector<JPtr> arr;
    std::vector<std::pair<std::string, JPtr>> obj;
    const JVal* get(const char* key) const {
        for (auto& kv : obj) if (kv.first == key) return kv.second.get();
        return nullptr;
    }
};

struct JParser {
    const char* p;
    const char* end;
    [[noreturn]] void fail(const char* what) const {
        throw std::runtime_error(std::string("JSON: ") + what + " at byte " + std::to_string((size_t)(p - begin)));
    }
    const char* begin;
    int depth = 0;
    JParser(const char* s, size_t n) : p(s), end(s + n), begin(s) {}
    void ws() { while (p < end && (*p == ' ' || *p == '\n' || *p == '\t' || *p == '\r')) ++p; }
    JPtr parse() {
        // recursive descent; the nesting depth of the reference's artifacts is a few hundred at most — bounded like the CBOR reader's,
        // so that 200 000 opening brackets are an error message and not a stack overflow (byte-level fuzz, round 6)
        struct Depth { int& d; explicit Depth(int& x) : d(x) { ++d; } ~Depth() { --d; } } guard(depth);
        if (depth > 4096) fail("nesting too deep");
        ws();
        if (p >= end) fail("unexpected end");
        JPtr v(new JVal());
        char c = *p;
        if (c == '{') {
            v->kind = JVal::Obj;
            ++p; ws();
            if (p < end && *p == '}') { ++p; return v; }
            for (;;) {
                ws();
                if (p >= end || *p != '"') fail("expected key");
                std::string k = str();
                ws();
                if (p >= end || *p != ':') fail("expected ':'");
                ++p;
                v->obj.emplace_back(std::move(k), parse());
                ws();
                if (p < end && *p == ',') { ++p; continue; }
                if (p < end && *p == '}') { ++p; break; }
                fail("expected ',' or '}'");
            }
        } else if (c == '[') {
            v->kind = JVal::Arr;
            ++p; ws();
            if (p < end && *p == ']') { ++p; return v; }
            for (;;) {
                v->arr.push_back(parse());
                ws();
                if (p < end && *p == ',') { ++p; continue; }
                if (p < end && *p == ']') { ++p; break; }
                fail("expected ',' or ']'");
            }
        } else if (c == '"') {
            v->kind = JVal::Str;
            v->s = str();
        } else if (c == '-' || (c >= '0' && c <= '9')) {
            v->kind = JVal::Num;
            if (c == '-') { v->neg = true; ++p; }
            if (p >= end || *p < '0' || *p > '9') fail("bad number");
            const char* num0 = p;
            bool wide = false;  // an integer beyond 64 bits: legal only as a float-like statistic, never as a field element or an index
            while (p < end && *p >= '0' && *p <= '9') {
                const uint64_t dgt = (uint64_t)(*p - '0');
                if (v->u > (UINT64_MAX - dgt) / 10) wide = true; else v->u = v->u * 10 + dgt;
                ++p;
            }
            if (wide && !(p < end && (*p == '.' || *p == 'e' || *p == 'E'))) fail("integer beyond 64 bits");
            if (p < end && (*p == '.' || *p == 'e' || *p == 'E')) {  // a float: only statistics carry them, never field elements
                while (p < end && ((*p >= '0' && *p <= '9') || *p == '.' || *p == 'e' || *p == 'E' || *p == '+' || *p == '-')) ++p;
                v->is_float = true;
                v->d = strtod(std::string(num0, p).c_str(), nullptr);
                if (v->neg) v->d = -v->d;
                v->u = 0;
            } else v->d = v->neg ? -(double)v->u : (double)v->u;
        } else if (!strncmp(p, "true", 4)) { v->kind = JVal::Bool; v->b = true; p += 4; }
        else if (!strncmp(p, "false", 5)) { v->kind = JVal::Bool; p += 5; }
        else if (!strncmp(p, "null", 4)) { p += 4; }
        else fail("unexpected character");
        return v;
    }
    std::string str() {
        ++p;  // opening quote
        std::string out;
        while (p < end && *p != '"') {
            if (*p == '\\') {
                ++p;
                if (p >= end) fail("bad escape");
                switch (*p) {
                    case 'n': out += '\n'; break;
                    case 't': out += '\t'; break;
                    case 'r': out += '\r'; break;
                    case 'b': out += '\b'; break;
                    case 'f': out += '\f'; break;
                    case 'u': {
                        if (end - p < 5) fail("bad \\u escape");
                        unsigned cp = (unsigned)strtoul(std::string(p + 1, p + 5).c_str(), nullptr, 16);
                        if (cp < 0x80) out += (char)cp;
                        else if (cp < 0x800) { out += (char)(0xC0 | (cp >> 6)); out += (char)(0x80 | (cp & 0x3F)); }
                        else { out += (char)(0xE0 | (cp >> 12)); out += (char)(0x80 | ((cp >> 6) & 0x3F)); out += (char)(0x80 | (cp & 0x3F)); }
                        p += 4;
                        break;
                    }
                    default: out += *p;
                }
                ++p;
            } else out += *p++;
        }
        if (p >= end) fail("unterminated string");
        ++p;
        return out;
    }
};

// ---------------------------------------------------------------------------------- CBOR
// RFC 8949 subset that serde_cbor emits for the reference's artifacts (cli-openvm-riscv/src/main.rs:380-407:
// `serde_cbor::to_writer(file, value)`): the same serde data model as the JSON export — structs are maps with text
// keys, sequences are arrays, newtype enum variants one-entry maps, unit variants text, Option::None null, field
// elements canonical unsigned integers — so it decodes into the same DOM and the same builders run on it.
// Indefinite-length items, tags (skipped), byte strings (kept as Str) and half/single/double floats are handled.
struct CborParser {
    const uint8_t* p;
    const uint8_t* end;
    const uint8_t* begin;
    int depth = 0;
    CborParser(const uint8_t* s, size_t n) : p(s), end(s + n), begin(s) {}
    [[noreturn]] void fail(const char* what) const {
        throw std::runtime_error(std::string("CBOR: ") + what + " at byte " + std::to_string((size_t)(p - begin)));
    }
    uint8_t byte() { if (p >= end) fail("unexpected end"); return *p++; }
    uint64_t arg(uint8_t info) {
        if (info < 24) return info;
        int n = info == 24 ? 1 : info == 25 ? 2 : info == 26 ? 4 : info == 27 ? 8 : 0;
        if (!n) fail("reserved additional information");
        uint64_t v = 0;
        for (int i = 0; i < n; ++i) v = (v << 8) | byte();
        return v;
    }
    static double half_to_double(uint16_t h) {
        const int e = (h >> 10) & 31, m = h & 1023;
        double v = e == 0 ? ldexp((double)m, -24) : e == 31 ? (m ? NAN : INFINITY) : ldexp((double)(m + 1024), e - 25);
        return (h & 0x8000) ? -v : v;
    }
    std::string text(uint8_t info, uint8_t major) {
        std::string out;
        if (info == 31) {  // indefinite: chunks until break
            for (;;) {
                const uint8_t b = byte();
                if (b == 0xff) break;
                if ((b >> 5) != major) fail("bad string chunk");
                const uint64_t n = arg(b & 31);
                if ((uint64_t)(end - p) < n) fail("string past the end");
                out.append((const char*)p, (size_t)n);
                p += n;
            }
            return out;
        }
        const uint64_t n = arg(info);
        if ((uint64_t)(end - p) < n) fail("string past the end");
        out.assign((const char*)p, (size_t)n);
        p += n;
        return out;
    }
    JPtr parse() {
        if (++depth > 4096) fail("nesting too deep");
        JPtr v(new JVal());
        uint8_t b = byte();
        while ((b >> 5) == 6) { (void)arg(b & 31); b = byte(); }  // tags: transparent
        const uint8_t major = b >> 5, info = b & 31;
        switch (major) {
            case 0: v->kind = JVal::Num; v->u = arg(info); v->d = (double)v->u; break;
            case 1: v->kind = JVal::Num; v->u = arg(info) + 1; v->neg = true; v->d = -(double)v->u; break;  // -1 - n
            case 2: case 3: v->kind = JVal::Str; v->s = text(info, major); break;
            case 4:
                v->kind = JVal::Arr;
                if (info == 31) { while (p < end && *p != 0xff) v->arr.push_back(parse()); (void)byte(); }
                else { const uint64_t n = arg(info); if (n > (uint64_t)(end - p)) fail("array longer than the input"); v->arr.reserve((size_t)n); for (uint64_t i = 0; i < n; ++i) v->arr.push_back(parse()); }
                break;
            case 5: {
                v->kind = JVal::Obj;
                auto entry = [&] {
                    JPtr k = parse();
                    std::string key = k->kind == JVal::Str ? k->s : k->kind == JVal::Num ? std::to_string(k->u) : std::string();
                    if (k->kind != JVal::Str && k->kind != JVal::Num) fail("unsupported map key type");
                    v->obj.emplace_back(std::move(key), parse());
                };
                if (info == 31) { while (p < end && *p != 0xff) entry(); (void)byte(); }
                else { const uint64_t n = arg(info); if (n > (uint64_t)(end - p)) fail("map longer than the input"); for (uint64_t i = 0; i < n; ++i) entry(); }
                break;
            }
            default:  // 7: simple values and floats
                if (info == 20 || info == 21) { v->kind = JVal::Bool; v->b = info == 21; }
                else if (info == 22 || info == 23) { /* null / undefined */ }
                else if (info == 25) { v->kind = JVal::Num; v->is_float = true; v->d = half_to_double((uint16_t)arg(info)); }
                else if (info == 26) { uint32_t w = (uint32_t)arg(info); float f; memcpy(&f, &w, 4); v->kind = JVal::Num; v->is_float = true; v->d = f; }
                else if (info == 27) { uint64_t w = arg(info); double d; memcpy(&d, &w, 8); v->kind = JVal::Num; v->is_float = true; v->d = d; }
                else if (info == 24) { (void)byte(); }
                else if (info < 20) { /* unassigned simple value: null */ }
                else fail("unexpected break / reserved simple value");
        }
        --depth;
        return v;
    }
};

// Every map of a document that looks like an `Apc` (keys block, machine, subs), in document order: the export files
// carry one at the top level (flattened into ApcWithBusMap, autoprecompiles/src/export.rs:271-276), the CLI's `select`
// artifact is a sequence of ApcWithStats{apc, stats, evaluation_result} (adapter.rs:22-27), the `setup` artifact nests
// them inside the VM configuration's PowdrExtension.
void find_apcs(const JVal& v, std::vector<const JVal*>& out) {
    std::vector<const JVal*> st{&v};
    while (!st.empty()) {
        const JVal* x = st.back();
        st.pop_back();
        if (x->kind == JVal::Obj) {
            const JVal* m = x->get("machine");
            if (x->get("block") && x->get("subs") && m && m->kind == JVal::Obj && m->get("constraints")) { out.push_back(x); continue; }
            for (auto it = x->obj.rbegin(); it != x->obj.rend(); ++it) st.push_back(it->second.get());
        } else if (x->kind == JVal::Arr) {
            for (auto it = x->arr.rbegin(); it != x->arr.rend(); ++it) st.push_back(it->get());
        }
    }
}

// ---------------------------------------------------------------------------------- model
enum NodeKind : uint8_t { N_NUM, N_REF, N_ADD, N_SUB, N_MUL, N_NEG };
struct Node { NodeKind kind; uint32_t a, b; };  // NUM: a = canonical value; REF: a = index into refs; NEG: a = child
struct BusInteraction { uint32_t id; uint32_t mult; std::vector<uint32_t> args; };
struct Derived { uint64_t poly_id; bool is_const; uint32_t constant; uint32_t e1, e2; };
struct Sub { uint32_t original_poly_index; uint64_t apc_poly_id; };

}  // namespace

struct PowdrApc {
    std::vector<Node> nodes;
    std::vector<uint64_t> ref_ids;                     // poly id per REF slot
    std::vector<uint32_t> constraints;                 // root node per constraint
    std::vector<BusInteraction> buses;
    std::vector<Derived> derived;
    std::vector<std::vector<uint32_t>> instructions;   // [opcode, a..g]
    std::vector<uint64_t> instr_pc;                    // pc of each instruction (block start_pc + 4 j; several blocks: a superblock)
    std::vector<std::vector<Sub>> subs;
    std::vector<uint64_t> poly_ids;                    // ascending
    std::unordered_map<uint64_t, uint32_t> id_to_index;
    // bus_map of an ApcWithBusMap export (autoprecompiles/src/bus_map.rs: {"bus_ids": {id: type}}); empty if absent
    struct BusMapEntry { uint64_t id; uint32_t kind; uint32_t sizes[2]; std::string name; };
    std::vector<BusMapEntry> bus_map;

    // device-side caches for generate_witness_gpu, keyed by trace height
    struct Compiled {
        uint32_t* d_dbc = nullptr; DerivedExprSpec* d_specs = nullptr; size_t n_specs = 0;
        uint32_t* d_bbc = nullptr; size_t bbc_len = 0; DevInteraction* d_inter = nullptr; size_t n_inter = 0;
        ExprSpan* d_spans = nullptr; size_t n_spans = 0;
        // host copies of the bus tables (shared, immutable): handed to powdr_apc_apply_bus_host_tables so that the replay
        // does not copy megabytes of bytecode back from the device to find its plan
        std::shared_ptr<const std::vector<uint32_t>> h_bbc;
        std::shared_ptr<const std::vector<DevInteraction>> h_inter;
        std::shared_ptr<const std::vector<ExprSpan>> h_spans;
    };
    std::map<std::pair<int, size_t>, Compiled> compiled;  // (device, height): one process may drive several GPUs
    // substitution tables keyed by the instr_air assignment hash (host only: the gather takes them from the host)
    struct SubTables { std::vector<Subst> subs; std::vector<int32_t> air_ids, row_block; };
    std::map<uint64_t, SubTables> sub_tables;
    // OriginalAir tables reach the gather as kernel arguments (powdr_apc_tracegen_host_tables, <= 16 AIRs). Only an APC
    // gathered from more AIRs than that needs a device copy: keyed by (device, content), immutable once uploaded, held by
    // shared_ptr so that a launch keeps its table alive while another host thread evicts it from the bounded cache.
    struct AirTable {
        std::vector<OriginalAir> h; OriginalAir* d = nullptr; int device = 0; uint64_t last_use = 0;
        ~AirTable() { if (d) (void)hipFree(d); }  // hipFree waits for the kernels already enqueued that read it
    };
    std::vector<std::shared_ptr<AirTable>> air_tables;
    // instruction table + record substitutions of the block (powdr_apc_generate_witness_from_records), built at first use
    struct RecordTables { std::vector<PowdrOrigInstr> instrs; std::vector<PowdrRecordSubst> subs; };
    std::shared_ptr<const RecordTables> record_tables;
    uint64_t air_clock = 0;
    std::mutex mu;  // guards the three caches (lookups and insertions; launches run outside it)

    ~PowdrApc() {
        for (auto& kv : compiled) {
            auto& c = kv.second;
            for (void* q : {(void*)c.d_dbc, (void*)c.d_specs, (void*)c.d_bbc, (void*)c.d_inter, (void*)c.d_spans}) if (q) (void)hipFree(q);
        }
    }
};

namespace {

uint64_t parse_ref(const std::string& s) {
    size_t pos = s.rfind('@');
    if (pos == std::string::npos) throw std::runtime_error("Invalid format for AlgebraicReference: " + s);
    char* e = nullptr;
    unsigned long long id = strtoull(s.c_str() + pos + 1, &e, 10);
    if (!e || *e != 0 || pos + 1 == s.size()) throw std::runtime_error("Invalid ID in AlgebraicReference: " + s.substr(pos + 1));
    return (uint64_t)id;
}

uint32_t field_of(const JVal& v) {
    uint64_t m = v.u % kP;
    return (uint32_t)(v.neg && m ? kP - m : m);
}

// expression/src/lib.rs:209-246: [l, op, r] | [op, e] | number | "name@id".
// Explicit work stack: the fixtures contain left-leaning sums thousands of terms deep.
uint32_t build_expr(PowdrApc& apc, const JVal& root) {
    struct Frame { const JVal* v; uint32_t slot; int state; uint32_t lhs; };
    auto leaf = [&](const JVal& v, uint32_t& out) -> bool {
        if (v.kind == JVal::Num) { apc.nodes.push_back({N_NUM, field_of(v), 0}); out = (uint32_t)apc.nodes.size() - 1; return true; }
        if (v.kind == JVal::Str) {
            apc.ref_ids.push_back(parse_ref(v.s));
            apc.nodes.push_back({N_REF, (uint32_t)apc.ref_ids.size() - 1, 0});
            out = (uint32_t)apc.nodes.size() - 1;
            return true;
        }
        return false;
    };
    uint32_t result = 0;
    if (leaf(root, result)) return result;
    std::vector<Frame> st;
    st.push_back({&root, 0, 0, 0});
    uint32_t ret = 0;
    while (!st.empty()) {
        Frame& f = st.back();
        const JVal& v = *f.v;
        if (v.kind != JVal::Arr || (v.arr.size() != 3 && v.arr.size() != 2)) throw std::runtime_error("cannot parse expression");
        const bool unary = v.arr.size() == 2;
        const JVal& first = unary ? *v.arr[1] : *v.arr[0];
        if (f.state == 0) {
            f.state = 1;
            uint32_t id;
            if (leaf(first, id)) { ret = id; } else { st.push_back({&first, 0, 0, 0}); continue; }
        }
        if (f.state == 1) {
            f.lhs = ret;
            if (unary) {
                if (v.arr[0]->kind != JVal::Str || v.arr[0]->s != "-") throw std::runtime_error("unknown unary operator");
                apc.nodes.push_back({N_NEG, f.lhs, 0});
                ret = (uint32_t)apc.nodes.size() - 1;
                st.pop_back();
                continue;
            }
            f.state = 2;
            uint32_t id;
            const JVal& second = *v.arr[2];
            if (leaf(second, id)) { ret = id; } else { st.push_back({&second, 0, 0, 0}); continue; }
        }
        // state 2: both operands ready (ret = rhs)
        {
            const JVal& op = *v.arr[1];
            if (op.kind != JVal::Str || op.s.size() != 1) throw std::runtime_error("bad binary operator");
            NodeKind k = op.s[0] == '+' ? N_ADD : op.s[0] == '-' ? N_SUB : op.s[0] == '*' ? N_MUL : N_NUM;
            if (k == N_NUM) throw std::runtime_error("unknown binary operator " + op.s);
            apc.nodes.push_back({k, f.lhs, ret});
            ret = (uint32_t)apc.nodes.size() - 1;
            st.pop_back();
        }
    }
    return ret;
}

// emit_expr, cuda/mod.rs:49-81 (post-order walk with an explicit stack)
void emit_expr(const PowdrApc& apc, uint32_t root, size_t apc_height, std::vector<uint32_t>& bc) {
    std::vector<std::pair<uint32_t, bool>> st;
    st.push_back({root, false});
    while (!st.empty()) {
        auto [n, done] = st.back();
        st.pop_back();
        const Node& nd = apc.nodes[n];
        switch (nd.kind) {
            case N_NUM: bc.push_back(POWDR_OP_PUSH_CONST); bc.push_back(nd.a); break;
            case N_REF: {
                uint32_t idx = apc.id_to_index.at(apc.ref_ids[nd.a]);
                bc.push_back(POWDR_OP_PUSH_APC);
                bc.push_back((uint32_t)((uint64_t)idx * apc_height));  // `as u32`, cuda/mod.rs:62
                break;
            }
            case N_NEG:
                if (done) bc.push_back(POWDR_OP_NEG);
                else { st.push_back({n, true}); st.push_back({nd.a, false}); }
                break;
            default:
                if (done) bc.push_back(nd.kind == N_ADD ? POWDR_OP_ADD : nd.kind == N_SUB ? POWDR_OP_SUB : POWDR_OP_MUL);
                else { st.push_back({n, true}); st.push_back({nd.b, false}); st.push_back({nd.a, false}); }
        }
    }
}

void collect_refs(const PowdrApc& apc, uint32_t root, std::vector<uint64_t>& out) {
    std::vector<uint32_t> st{root};
    while (!st.empty()) {
        uint32_t n = st.back(); st.pop_back();
        const Node& nd = apc.nodes[n];
        if (nd.kind == N_REF) out.push_back(apc.ref_ids[nd.a]);
        else if (nd.kind == N_NEG) st.push_back(nd.a);
        else if (nd.kind != N_NUM) { st.push_back(nd.a); st.push_back(nd.b); }
    }
}

uint64_t next_pow2_or_zero(uint64_t n) { if (!n) return 0; uint64_t p = 1; while (p < n) p <<= 1; return p; }

template <class T> int upload(T*& d, const std::vector<T>& h) {
    size_t bytes = (h.size() ? h.size() : 1) * sizeof(T);
    hipError_t e = hipMalloc((void**)&d, bytes);
    if (e != hipSuccess) return (int)e;
    if (!h.empty()) { e = hipMemcpy(d, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice); if (e != hipSuccess) return (int)e; }
    return 0;
}

struct BusTables { std::vector<DevInteraction> inter; std::vector<ExprSpan> spans; std::vector<uint32_t> bc; };
BusTables compile_bus(const PowdrApc& apc, size_t h) {
    BusTables t;
    for (auto& b : apc.buses) {
        uint32_t off_idx = (uint32_t)t.spans.size();
        auto span = [&](uint32_t e) { uint32_t off = (uint32_t)t.bc.size(); emit_expr(apc, e, h, t.bc); t.spans.push_back({off, (uint32_t)t.bc.size() - off}); };
        span(b.mult);
        for (uint32_t a : b.args) span(a);
        t.inter.push_back({b.id, (uint32_t)b.args.size(), off_idx});
    }
    return t;
}
struct DerivedTables { std::vector<DerivedExprSpec> specs; std::vector<uint32_t> bc; };
DerivedTables compile_derived(const PowdrApc& apc, size_t h) {
    DerivedTables t;
    for (auto& d : apc.derived) {
        uint32_t off = (uint32_t)t.bc.size();
        if (d.is_const) { t.bc.push_back(POWDR_OP_PUSH_CONST); t.bc.push_back(d.constant); }
        else {
            emit_expr(apc, d.e2, h, t.bc); t.bc.push_back(POWDR_OP_INV_OR_ZERO);
            emit_expr(apc, d.e1, h, t.bc); t.bc.push_back(POWDR_OP_MUL);
        }
        DerivedExprSpec s;
        s.col_base = (uint64_t)apc.id_to_index.at(d.poly_id) * h;
        s.span = {off, (uint32_t)t.bc.size() - off};
        t.specs.push_back(s);
    }
    return t;
}

uint64_t fnv(const void* p, size_t n) { uint64_t h = 1469598103934665603ull; auto c = (const unsigned char*)p; for (size_t i = 0; i < n; ++i) { h ^= c[i]; h *= 1099511628211ull; } return h; }

}  // namespace

namespace {

// bus_map: {"bus_ids": {"<id>": "ExecutionBridge" | "Memory" | "PcLookup" | {"Other": "VariableRangeChecker" | "BitwiseLookup" |
// {"TupleRangeChecker": [sz0, sz1]}}}} (autoprecompiles/src/bus_map.rs:4-16, openvm-bus-interaction-handler/src/bus_map.rs:17-21)
void parse_bus_map(PowdrApc& apc, const JVal& bm) {
    const JVal* ids = bm.get("bus_ids");
    if (!ids || ids->kind != JVal::Obj) return;
    for (auto& kv : ids->obj) {
        PowdrApc::BusMapEntry e{};
        e.id = strtoull(kv.first.c_str(), nullptr, 10);
        const JVal* t = kv.second.get();
        e.kind = POWDR_BUS_OTHER;
        auto named = [&](const std::string& n) {
            e.name = n;
            if (n == "ExecutionBridge") e.kind = POWDR_BUS_EXECUTION_BRIDGE;
            else if (n == "Memory") e.kind = POWDR_BUS_MEMORY;
            else if (n == "PcLookup") e.kind = POWDR_BUS_PC_LOOKUP;
            else if (n == "VariableRangeChecker") e.kind = POWDR_BUS_VARIABLE_RANGE_CHECKER;
            else if (n == "BitwiseLookup") e.kind = POWDR_BUS_BITWISE_LOOKUP;
            else if (n == "TupleRangeChecker") e.kind = POWDR_BUS_TUPLE_RANGE_CHECKER;
        };
        if (t->kind == JVal::Str) named(t->s);
        else if (t->kind == JVal::Obj && t->obj.size() == 1) {
            const JVal* o = t->obj[0].second.get();  // {"Other": ...}
            if (t->obj[0].first != "Other") { named(t->obj[0].first); o = nullptr; }
            if (o && o->kind == JVal::Str) named(o->s);
            else if (o && o->kind == JVal::Obj && o->obj.size() == 1) {
                named(o->obj[0].first);
                const JVal* sz = o->obj[0].second.get();
                if (sz->kind == JVal::Arr) for (size_t k = 0; k < sz->arr.size() && k < 2; ++k) e.sizes[k] = (uint32_t)sz->arr[k]->u;
            }
        }
        apc.bus_map.push_back(std::move(e));
    }
}

// One `Apc` from its DOM (JSON or CBOR): keys block, machine{constraints, bus_interactions, derived_columns}, subs [, bus_map]
std::unique_ptr<PowdrApc> build_apc_from_dom(const JVal& root) {
    std::unique_ptr<PowdrApc> apc(new PowdrApc());
    const JVal* block = root.get("block");
    const JVal* machine = root.get("machine");
    const JVal* subs = root.get("subs");
    if (!block || !machine || !subs) throw std::runtime_error("missing block/machine/subs");
    const JVal* blocks = block->get("blocks");
    if (!blocks) throw std::runtime_error("missing block.blocks");
    for (auto& b : blocks->arr) {
        const JVal* ins = b->get("instructions");
        if (!ins) throw std::runtime_error("missing instructions");
        const JVal* spc = b->get("start_pc");
        uint64_t pc = spc && spc->kind == JVal::Num ? spc->u : 0;
        for (auto& i : ins->arr) {
            apc->instr_pc.push_back(pc);
            pc += 4;
            std::vector<uint32_t> v;
            if (i->kind == JVal::Arr) {  // export files: SimpleInstruction = [opcode, a, b, c, d, e, f, g] (export.rs:221-251)
                for (auto& x : i->arr) v.push_back((uint32_t)x->u);
            } else if (i->kind == JVal::Obj) {  // serde of the VM's own instruction struct: {"opcode": n, "a": .., ...}
                const JVal* ins_obj = i.get();
                while (ins_obj->kind == JVal::Obj && ins_obj->obj.size() == 1 && !ins_obj->get("opcode")) ins_obj = ins_obj->obj[0].second.get();  // newtype wrappers
                const JVal* op = ins_obj->kind == JVal::Obj ? ins_obj->get("opcode") : nullptr;
                if (!op) throw std::runtime_error("instruction without an opcode");
                v.push_back((uint32_t)op->u);
                for (const char* f : {"a", "b", "c", "d", "e", "f", "g"}) if (const JVal* x = ins_obj->get(f)) v.push_back(x->kind == JVal::Num ? (uint32_t)x->u : 0u);
            } else throw std::runtime_error("cannot parse instruction");
            apc->instructions.push_back(std::move(v));
        }
    }
    for (auto& row : subs->arr) {
        std::vector<Sub> r;
        for (auto& s : row->arr) {
            const JVal* o = s->get("original_poly_index");
            const JVal* a = s->get("apc_poly_id");
            if (!o || !a) throw std::runtime_error("bad substitution");
            r.push_back({(uint32_t)o->u, a->u});
        }
        apc->subs.push_back(std::move(r));
    }
    if (apc->subs.size() != apc->instructions.size()) throw std::runtime_error("subs / instructions length mismatch (zip_eq)");
    const JVal* cons = machine->get("constraints");
    const JVal* buses = machine->get("bus_interactions");
    const JVal* derived = machine->get("derived_columns");
    if (!cons || !buses || !derived) throw std::runtime_error("missing machine fields");
    for (auto& c : cons->arr) apc->constraints.push_back(build_expr(*apc, *c));
    for (auto& b : buses->arr) {
        BusInteraction bi;
        const JVal* id = b->get("id"); const JVal* m = b->get("mult"); const JVal* args = b->get("args");
        if (!id || !m || !args) throw std::runtime_error("bad bus interaction");
        bi.id = (uint32_t)id->u;
        bi.mult = build_expr(*apc, *m);
        for (auto& a : args->arr) bi.args.push_back(build_expr(*apc, *a));
        apc->buses.push_back(std::move(bi));
    }
    for (auto& d : derived->arr) {
        if (d->kind != JVal::Arr || d->arr.size() != 2) throw std::runtime_error("bad derived column");
        Derived dv{};
        dv.poly_id = parse_ref(d->arr[0]->s);
        const JVal& method = *d->arr[1];
        if (const JVal* c = method.get("Constant")) { dv.is_const = true; dv.constant = field_of(*c); }
        else if (const JVal* q = method.get("QuotientOrZero")) {
            if (q->arr.size() != 2) throw std::runtime_error("bad QuotientOrZero");
            dv.e1 = build_expr(*apc, *q->arr[0]);
            dv.e2 = build_expr(*apc, *q->arr[1]);
        } else throw std::runtime_error("unknown ComputationMethod");
        apc->derived.push_back(dv);
    }
    // main_columns: unique references in constraints and bus interactions, ascending id
    std::vector<uint64_t> ids;
    for (uint32_t c : apc->constraints) collect_refs(*apc, c, ids);
    for (auto& b : apc->buses) { collect_refs(*apc, b.mult, ids); for (uint32_t a : b.args) collect_refs(*apc, a, ids); }
    std::sort(ids.begin(), ids.end());
    ids.erase(std::unique(ids.begin(), ids.end()), ids.end());
    apc->poly_ids = ids;
    for (size_t i = 0; i < ids.size(); ++i) apc->id_to_index[ids[i]] = (uint32_t)i;
    // every substitution / derived target must be a main column (the reference indexes the BTreeMap)
    for (auto& row : apc->subs) for (auto& s : row) if (!apc->id_to_index.count(s.apc_poly_id)) throw std::runtime_error("substitution targets an unknown column");
    for (auto& d : apc->derived) if (!apc->id_to_index.count(d.poly_id)) throw std::runtime_error("derived column is not a main column");
    if (const JVal* bm = root.get("bus_map")) parse_bus_map(*apc, *bm);
    return apc;
}

PowdrApc* apc_at(const JVal& root, size_t index, char* err, size_t err_cap) {
    try {
        std::vector<const JVal*> found;
        find_apcs(root, found);
        if (found.empty()) throw std::runtime_error("missing block/machine/subs: no Apc in the document");
        if (index >= found.size()) throw std::runtime_error("Apc index " + std::to_string(index) + " out of range (" + std::to_string(found.size()) + " in the document)");
        return build_apc_from_dom(*found[index]).release();
    } catch (const std::exception& e) {
        if (err && err_cap) snprintf(err, err_cap, "%s", e.what());
        return nullptr;
    }
}

template <class F> size_t count_apcs(F parse_root) {
    try {
        JPtr root = parse_root();
        std::vector<const JVal*> found;
        find_apcs(*root, found);
        return found.size();
    } catch (const std::exception&) {
        return 0;
    }
}

}  // namespace

extern "C" {

PowdrApc* powdr_apc_from_json_at(const char* json, size_t len, size_t index, char* err, size_t err_cap) {
    JPtr root;
    try {
        JParser jp(json, len);
        root = jp.parse();
    } catch (const std::exception& e) {
        if (err && err_cap) snprintf(err, err_cap, "%s", e.what());
        return nullptr;
    }
    return apc_at(*root, index, err, err_cap);
}
PowdrApc* powdr_apc_from_json(const char* json, size_t len, char* err, size_t err_cap) { return powdr_apc_from_json_at(json, len, 0, err, err_cap); }
size_t powdr_apc_count_in_json(const char* json, size_t len) { return count_apcs([&] { JParser jp(json, len); return jp.parse(); }); }

PowdrApc* powdr_apc_from_cbor(const uint8_t* bytes, size_t len, size_t index, char* err, size_t err_cap) {
    JPtr root;
    try {
        CborParser cp(bytes, len);
        root = cp.parse();
    } catch (const std::exception& e) {
        if (err && err_cap) snprintf(err, err_cap, "%s", e.what());
        return nullptr;
    }
    return apc_at(*root, index, err, err_cap);
}
size_t powdr_apc_count_in_cbor(const uint8_t* bytes, size_t len) { return count_apcs([&] { CborParser cp(bytes, len); return cp.parse(); }); }

size_t powdr_apc_bus_map_len(const PowdrApc* apc) { return apc->bus_map.size(); }
int powdr_apc_bus_map_entry(const PowdrApc* apc, size_t i, uint64_t* bus_id, uint32_t* kind, uint32_t* sizes2, char* name, size_t name_cap) {
    if (i >= apc->bus_map.size()) return -1;
    const auto& e = apc->bus_map[i];
    if (bus_id) *bus_id = e.id;
    if (kind) *kind = e.kind;
    if (sizes2) { sizes2[0] = e.sizes[0]; sizes2[1] = e.sizes[1]; }
    if (name && name_cap) snprintf(name, name_cap, "%s", e.name.c_str());
    return 0;
}
int powdr_apc_periphery_from_bus_map(const PowdrApc* apc, PowdrPeriphery* per) {
    int found = 0;
    for (const auto& e : apc->bus_map) {
        if (e.kind == POWDR_BUS_VARIABLE_RANGE_CHECKER) { per->var_range_bus_id = (uint32_t)e.id; ++found; }
        else if (e.kind == POWDR_BUS_BITWISE_LOOKUP) { per->bitwise_bus_id = (uint32_t)e.id; ++found; }
        else if (e.kind == POWDR_BUS_TUPLE_RANGE_CHECKER) { per->tuple2_bus_id = (uint32_t)e.id; per->tuple2_sz0 = e.sizes[0]; per->tuple2_sz1 = e.sizes[1]; ++found; }
    }
    return found;
}

// ---- apc_candidates.json (autoprecompiles/src/pgo/cell/mod.rs:34-97) ----
struct PowdrApcCandidates {
    uint64_t version = 0;
    std::vector<PowdrApcCandidateInfo> apcs;
    size_t n_labels = 0;
};

PowdrApcCandidates* powdr_apc_candidates_from_json(const char* json, size_t len, char* err, size_t err_cap) {
    std::unique_ptr<PowdrApcCandidates> out(new PowdrApcCandidates());
    try {
        JParser jp(json, len);
        JPtr root = jp.parse();
        const JVal* apcs = root->kind == JVal::Arr ? root.get() : root->get("apcs");  // version 0 was a bare array
        if (const JVal* v = root->get("version")) out->version = v->u;
        if (!apcs || apcs->kind != JVal::Arr) throw std::runtime_error("missing apcs");
        if (const JVal* l = root->get("labels")) out->n_labels = l->obj.size();
        auto stats = [](const JVal* s, PowdrAirStats& o) {
            if (!s) throw std::runtime_error("missing stats");
            const JVal* m = s->get("main_columns"); const JVal* c = s->get("constraints"); const JVal* b = s->get("bus_interactions");
            if (!m || !c || !b) throw std::runtime_error("bad AirStats");
            o.main_columns = m->u; o.constraints = c->u; o.bus_interactions = b->u;
        };
        for (auto& a : apcs->arr) {
            PowdrApcCandidateInfo ci{};
            const JVal* f = a->get("execution_frequency");
            const JVal* st = a->get("stats");
            if (!f || !st) throw std::runtime_error("bad candidate");
            ci.execution_frequency = f->u;
            stats(st->get("before"), ci.before);
            stats(st->get("after"), ci.after);
            // version >= 4: original_blocks (superblocks); 2..3: original_block; each {start_pc, instructions}
            const JVal* blocks = a->get("original_blocks");
            std::vector<const JVal*> bl;
            if (blocks && blocks->kind == JVal::Arr) for (auto& b : blocks->arr) bl.push_back(b.get());
            else if (const JVal* b1 = a->get("original_block")) bl.push_back(b1);
            ci.n_blocks = (uint32_t)bl.size();
            for (size_t k = 0; k < bl.size(); ++k) {
                if (const JVal* pc = bl[k]->get("start_pc")) { if (k == 0) ci.start_pc = pc->u; }
                const JVal* ins = bl[k]->get("instructions");
                if (!ins) ins = bl[k]->get("statements");  // version < 2
                if (ins) ci.n_instructions += (uint32_t)ins->arr.size();
            }
            if (const JVal* w = a->get("width_before")) ci.width_before = w->u;
            if (const JVal* v = a->get("value")) ci.value = v->u;
            if (const JVal* c = a->get("cost_before")) ci.cost_before = c->d;
            if (const JVal* c = a->get("cost_after")) ci.cost_after = c->d;
            out->apcs.push_back(ci);
        }
    } catch (const std::exception& e) {
        if (err && err_cap) snprintf(err, err_cap, "%s", e.what());
        return nullptr;
    }
    return out.release();
}
void powdr_apc_candidates_free(PowdrApcCandidates* c) { delete c; }
uint64_t powdr_apc_candidates_version(const PowdrApcCandidates* c) { return c->version; }
size_t powdr_apc_candidates_count(const PowdrApcCandidates* c) { return c->apcs.size(); }
size_t powdr_apc_candidates_num_labels(const PowdrApcCandidates* c) { return c->n_labels; }
int powdr_apc_candidates_get(const PowdrApcCandidates* c, size_t i, PowdrApcCandidateInfo* out) {
    if (i >= c->apcs.size() || !out) return -1;
    *out = c->apcs[i];
    return 0;
}

void powdr_apc_free(PowdrApc* apc) { delete apc; }
uint32_t powdr_apc_width(const PowdrApc* a) { return (uint32_t)a->poly_ids.size(); }
const uint64_t* powdr_apc_poly_ids(const PowdrApc* a) { return a->poly_ids.data(); }
uint32_t powdr_apc_num_constraints(const PowdrApc* a) { return (uint32_t)a->constraints.size(); }
uint32_t powdr_apc_num_bus_interactions(const PowdrApc* a) { return (uint32_t)a->buses.size(); }
uint32_t powdr_apc_num_derived_columns(const PowdrApc* a) { return (uint32_t)a->derived.size(); }
uint32_t powdr_apc_num_instructions(const PowdrApc* a) { return (uint32_t)a->instructions.size(); }
uint32_t powdr_apc_instruction_opcode(const PowdrApc* a, uint32_t i) { return a->instructions[i].empty() ? 0 : a->instructions[i][0]; }
uint32_t powdr_apc_instruction_num_subs(const PowdrApc* a, uint32_t i) { return (uint32_t)a->subs[i].size(); }

size_t powdr_apc_compile_bus(const PowdrApc* apc, size_t h, DevInteraction* inter, ExprSpan* spans, size_t* n_spans, uint32_t* bc) {
    BusTables t = compile_bus(*apc, h);
    if (n_spans) *n_spans = t.spans.size();
    if (inter && !t.inter.empty()) memcpy(inter, t.inter.data(), t.inter.size() * sizeof(DevInteraction));
    if (spans && !t.spans.empty()) memcpy(spans, t.spans.data(), t.spans.size() * sizeof(ExprSpan));
    if (bc && !t.bc.empty()) memcpy(bc, t.bc.data(), t.bc.size() * 4);  // (an empty vector's data() may be null: UBSan, round 6)
    return t.bc.size();
}

size_t powdr_apc_compile_derived(const PowdrApc* apc, size_t h, DerivedExprSpec* specs, uint32_t* bc) {
    DerivedTables t = compile_derived(*apc, h);
    if (specs && !t.specs.empty()) memcpy(specs, t.specs.data(), t.specs.size() * sizeof(DerivedExprSpec));
    if (bc && !t.bc.empty()) memcpy(bc, t.bc.data(), t.bc.size() * 4);
    return t.bc.size();
}

size_t powdr_apc_compile_constraints(const PowdrApc* apc, ExprSpan* spans, uint32_t* bc) {
    std::vector<uint32_t> out;
    size_t k = 0;
    for (uint32_t c : apc->constraints) {
        uint32_t off = (uint32_t)out.size();
        emit_expr(*apc, c, 1, out);
        if (spans) spans[k] = {off, (uint32_t)out.size() - off};
        ++k;
    }
    if (bc && !out.empty()) memcpy(bc, out.data(), out.size() * 4);
    return out.size();
}

size_t powdr_apc_build_substitutions(const PowdrApc* apc, const int32_t* instr_air, Subst* subs, int32_t* air_ids_out,
                                     int32_t* row_block_out, size_t* n_airs) {
    // group instructions (with substitutions) by AIR, in order of first appearance
    std::vector<int32_t> order;
    std::unordered_map<int32_t, size_t> slot;
    std::vector<std::vector<uint32_t>> rows;  // instruction indices per AIR slot
    for (size_t i = 0; i < apc->instructions.size(); ++i) {
        if (apc->subs[i].empty()) continue;
        int32_t a = instr_air[i];
        auto it = slot.find(a);
        if (it == slot.end()) { it = slot.emplace(a, order.size()).first; order.push_back(a); rows.emplace_back(); }
        rows[it->second].push_back((uint32_t)i);
    }
    size_t n = 0;
    for (size_t k = 0; k < order.size(); ++k) {
        for (size_t row = 0; row < rows[k].size(); ++row)
            for (auto& s : apc->subs[rows[k][row]]) {
                if (subs) subs[n] = {(int32_t)k, (int32_t)s.original_poly_index, (int32_t)row, (int32_t)apc->id_to_index.at(s.apc_poly_id)};
                ++n;
            }
        if (air_ids_out) air_ids_out[k] = order[k];
        if (row_block_out) row_block_out[k] = (int32_t)rows[k].size();
    }
    if (n_airs) *n_airs = order.size();
    return n;
}

}  // extern "C"

namespace {

// cuda/mod.rs:266-269 zero-fills the whole matrix so that columns covered by neither a substitution nor a derived expression
// read as zero. The gather and the derived-column kernel write EVERY row of the columns they cover (padding rows included), so
// only the uncovered columns are cleared here (normally none: at C2 this saves an 8.5 GB memset per segment).
int clear_uncovered_columns(const PowdrApc* apc, PowdrFp* d_output, size_t height) {
    const size_t width = apc->poly_ids.size();
    std::vector<char> covered(width, 0);
    for (auto& row : apc->subs) for (auto& s : row) covered[apc->id_to_index.at(s.apc_poly_id)] = 1;
    for (auto& d : apc->derived) covered[apc->id_to_index.at(d.poly_id)] = 1;
    for (size_t c = 0; c < width;) {
        if (covered[c]) { ++c; continue; }
        size_t e = c;
        while (e < width && !covered[e]) ++e;
        PW_HIP_TRY(hipMemsetAsync(d_output + c * height, 0, (e - c) * height * sizeof(PowdrFp), pw::stream()));
        c = e;
    }
    return 0;
}

int apply_derived_and_bus(PowdrApc* apc, size_t height, size_t num_calls, PowdrFp* d_output, const PowdrPeriphery* per);

}  // namespace

extern "C" {

int powdr_apc_generate_witness_gpu(PowdrApc* apc, const int32_t* instr_air, const PowdrDeviceMatrix* dummy, size_t n_dummy,
                                   size_t num_calls, PowdrFp* d_output, const PowdrPeriphery* per) {
    const size_t height = (size_t)next_pow2_or_zero(num_calls);
    if (height == 0) return 0;  // the APC was not called: DeviceMatrix::dummy()
    if (int rc0 = clear_uncovered_columns(apc, d_output, height)) return rc0;

    // ---- OriginalAir / Subst tables (cuda/mod.rs:272-332) ----
    int rc = 0;
    int device = 0;
    PW_HIP_TRY(hipGetDevice(&device));
    const PowdrApc::SubTables* stb_p = nullptr;
    std::shared_ptr<PowdrApc::AirTable> air_table;  // only for more than 16 AIRs; alive until the launch is enqueued
    std::vector<OriginalAir> airs;
    {
        std::lock_guard<std::mutex> lk(apc->mu);
        uint64_t key = fnv(instr_air, apc->instructions.size() * sizeof(int32_t));
        auto it = apc->sub_tables.find(key);
        if (it == apc->sub_tables.end()) {
            PowdrApc::SubTables t;
            size_t n_airs = 0;
            size_t n = powdr_apc_build_substitutions(apc, instr_air, nullptr, nullptr, nullptr, &n_airs);
            t.subs.resize(n); t.air_ids.resize(n_airs); t.row_block.resize(n_airs);
            powdr_apc_build_substitutions(apc, instr_air, t.subs.data(), t.air_ids.data(), t.row_block.data(), &n_airs);
            it = apc->sub_tables.emplace(key, std::move(t)).first;
        }
        stb_p = &it->second;  // std::map nodes are stable; entries are never erased
        const PowdrApc::SubTables& stb = *stb_p;
        airs.resize(stb.air_ids.size());
        for (size_t k = 0; k < airs.size(); ++k) {
            int32_t id = stb.air_ids[k];
            if (id < 0 || (size_t)id >= n_dummy) return -1;
            memset(&airs[k], 0, sizeof(OriginalAir));  // padding bytes take part in the content compare
            airs[k].width = dummy[id].width; airs[k].height = dummy[id].height; airs[k].buffer = dummy[id].buffer;
            airs[k].row_block_size = stb.row_block[k];
        }
        if (airs.size() > 16) {
            for (auto& t : apc->air_tables)
                if (t->device == device && t->h.size() == airs.size() && memcmp(t->h.data(), airs.data(), airs.size() * sizeof(OriginalAir)) == 0) { air_table = t; break; }
            if (!air_table) {
                if (apc->air_tables.size() >= 16) {  // bounded: drop the least recently used table (a launch in flight holds its own reference)
                    size_t v = 0;
                    for (size_t k = 1; k < apc->air_tables.size(); ++k) if (apc->air_tables[k]->last_use < apc->air_tables[v]->last_use) v = k;
                    apc->air_tables.erase(apc->air_tables.begin() + (long)v);
                }
                air_table = std::make_shared<PowdrApc::AirTable>();
                air_table->h = airs;
                air_table->device = device;
                PW_HIP_TRY(hipMalloc((void**)&air_table->d, airs.size() * sizeof(OriginalAir)));
                PW_HIP_TRY(hipMemcpy(air_table->d, airs.data(), airs.size() * sizeof(OriginalAir), hipMemcpyHostToDevice));
                apc->air_tables.push_back(air_table);
            }
            air_table->last_use = ++apc->air_clock;
        }
    }
    const PowdrApc::SubTables& stb = *stb_p;
    // host copies of both tables are at hand: no device-to-host round trip to find the gather plan, and (<= 16 AIRs) no
    // device table at all — the records travel as kernel arguments
    rc = powdr_apc_tracegen_host_tables(d_output, height, air_table ? air_table->d : nullptr, airs.data(), airs.size(), stb.subs.data(),
                                        stb.subs.size(), (int)num_calls);
    if (rc) return rc;
    return apply_derived_and_bus(apc, height, num_calls, d_output, per);
}

// The instruction table of the APC's block for the record expanders (include/powdr_gpu.h): the instructions that keep at least
// one cell (cuda/mod.rs:283-291 drops the others) in program order, with their pcs, timestamp offsets (every instruction of the
// block advances the timestamp by its accesses, kept or not), rows inside their AIR's block and record offsets. Returns their
// number ((size_t)-1: an opcode outside the thirteen chips); *words_per_call = size of one call's record.
size_t powdr_apc_instruction_table(const PowdrApc* apc, PowdrOrigInstr* out, size_t* words_per_call) {
    uint32_t air_rows[orig::kKinds] = {};
    uint32_t rec_off = 1, ts = 0;  // record word 0 = the call's first timestamp
    size_t n = 0;
    for (size_t i = 0; i < apc->instructions.size(); ++i) {
        const auto& ins = apc->instructions[i];
        const int k = ins.empty() ? -1 : orig::kind_of_opcode(ins[0]);
        if (k < 0) return (size_t)-1;
        if (!apc->subs[i].empty()) {
            if (out) {
                auto f = [&](size_t j) { return j < ins.size() ? ins[j] : 0u; };
                out[n] = PowdrOrigInstr{(uint32_t)k, ins[0], (uint32_t)apc->instr_pc[i], f(1), f(2), f(3), f(5), f(6), f(7), ts, air_rows[k], rec_off};
            }
            ++n;
            ++air_rows[k];
            rec_off += (uint32_t)orig::kRecordWords[k];
        }
        ts += (uint32_t)orig::kAccesses[k];
    }
    if (words_per_call) *words_per_call = rec_off;
    return n;
}

// try_generate_witness (cuda/mod.rs:201-401) with the original chips' work folded in: the APC trace straight from the call
// records (powdr_apc_tracegen_records: no dummy traces), then derived columns and bus replay as above.
int powdr_apc_generate_witness_from_records(PowdrApc* apc, const uint32_t* d_records, size_t num_calls, PowdrFp* d_output,
                                            const PowdrPeriphery* per) {
    const size_t height = (size_t)next_pow2_or_zero(num_calls);
    if (height == 0) return 0;
    if (int rc0 = clear_uncovered_columns(apc, d_output, height)) return rc0;
    std::shared_ptr<const PowdrApc::RecordTables> rt;
    {
        std::lock_guard<std::mutex> lk(apc->mu);
        if (!apc->record_tables) {
            auto t = std::make_shared<PowdrApc::RecordTables>();
            const size_t n = powdr_apc_instruction_table(apc, nullptr, nullptr);
            if (n == (size_t)-1) return (int)hipErrorInvalidValue;
            t->instrs.resize(n);
            powdr_apc_instruction_table(apc, t->instrs.data(), nullptr);
            int32_t k = 0;
            for (size_t i = 0; i < apc->instructions.size(); ++i) {
                if (apc->subs[i].empty()) continue;
                for (auto& s : apc->subs[i]) t->subs.push_back(PowdrRecordSubst{k, (int32_t)s.original_poly_index, (int32_t)apc->id_to_index.at(s.apc_poly_id)});
                ++k;
            }
            apc->record_tables = t;
        }
        rt = apc->record_tables;
    }
    int rc = powdr_apc_tracegen_records(d_output, height, d_records, num_calls, rt->instrs.data(), rt->instrs.size(), rt->subs.data(), rt->subs.size());
    if (rc) return rc;
    return apply_derived_and_bus(apc, height, num_calls, d_output, per);
}

}  // extern "C"

namespace {

int apply_derived_and_bus(PowdrApc* apc, size_t height, size_t num_calls, PowdrFp* d_output, const PowdrPeriphery* per) {
    const size_t width = apc->poly_ids.size();
    int rc = 0;
    int device = 0;
    PW_HIP_TRY(hipGetDevice(&device));
    // ---- derived columns + bus interactions, compiled once per height ----
    // Traces with width*height >= 2^32 cannot be addressed by the reference's u32 element offsets:
    // compile with column-index operands and use the *_cols entry points instead.
    const bool wide = (uint64_t)width * (uint64_t)height > 0xffffffffull;
    std::unique_lock<std::mutex> lk(apc->mu);
    auto ct = apc->compiled.find({device, height});
    if (ct == apc->compiled.end()) {
        PowdrApc::Compiled c;
        DerivedTables d = compile_derived(*apc, wide ? 1 : height);
        if (wide) for (auto& sp : d.specs) sp.col_base *= height;
        BusTables b = compile_bus(*apc, wide ? 1 : height);
        c.n_specs = d.specs.size(); c.bbc_len = b.bc.size(); c.n_inter = b.inter.size(); c.n_spans = b.spans.size();
        if ((rc = upload(c.d_specs, d.specs)) || (rc = upload(c.d_dbc, d.bc)) || (rc = upload(c.d_bbc, b.bc)) ||
            (rc = upload(c.d_inter, b.inter)) || (rc = upload(c.d_spans, b.spans))) return rc;
        c.h_bbc = std::make_shared<const std::vector<uint32_t>>(std::move(b.bc));
        c.h_inter = std::make_shared<const std::vector<DevInteraction>>(std::move(b.inter));
        c.h_spans = std::make_shared<const std::vector<ExprSpan>>(std::move(b.spans));
        ct = apc->compiled.emplace(std::make_pair(device, height), c).first;
    }
    const PowdrApc::Compiled c = ct->second;  // device tables are immutable once uploaded
    lk.unlock();
    rc = wide ? powdr_apc_apply_derived_expr_cols(d_output, height, (int)num_calls, c.d_specs, c.n_specs, c.d_dbc)
              : _apc_apply_derived_expr(d_output, height, (int)num_calls, c.d_specs, c.n_specs, c.d_dbc);
    if (rc) return rc;
    if (per) {
        rc = powdr_apc_apply_bus_host_tables(d_output, wide ? height : 0, (int)num_calls, c.d_bbc, c.h_bbc->data(), c.bbc_len, c.d_inter,
                                             c.h_inter->data(), c.n_inter, c.d_spans, c.h_spans->data(), c.n_spans,
                                             per->var_range_bus_id, per->d_var_hist, per->var_num_bins, per->tuple2_bus_id,
                                             per->d_tuple2_hist, per->tuple2_sz0, per->tuple2_sz1, per->bitwise_bus_id,
                                             per->d_bitwise_hist);
        if (rc) return rc;
    }
    return 0;
}

}  // namespace
