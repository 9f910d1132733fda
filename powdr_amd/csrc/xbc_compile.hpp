// Host-side compiler: reference post-fix bytecode -> xbc (see xbc.hpp).
#pragma once
#include "xbc.hpp"
#include "../../include/powdr_gpu.h"

#include <vector>

namespace xbc {

struct Node {
    enum Kind : uint8_t { COL, CONST, ADD, SUB, MUL, NEG, INV } kind;
    uint32_t a;  // COL: operand; CONST: canonical value; unary: child; binary: left
    uint32_t b;  // binary: right
    uint8_t need;
};

class Compiler {
public:
    // Compile one post-fix expression. Returns false if the code is malformed (stack underflow, leftovers,
    // unknown opcode, truncated operand) or needs more than `max_depth` stack slots.
    bool compile(const uint32_t* bc, uint32_t len, std::vector<uint32_t>& out, int max_depth = POWDR_EXPR_STACK_CAPACITY - 1) {
        nodes_.clear();
        std::vector<uint32_t> st;
        for (uint32_t ip = 0; ip < len;) {
            const uint32_t op = bc[ip++];
            switch (op) {
                case POWDR_OP_PUSH_APC:
                case POWDR_OP_PUSH_CONST:
                    if (ip >= len) return false;
                    st.push_back(leaf(op == POWDR_OP_PUSH_APC ? Node::COL : Node::CONST,
                                      op == POWDR_OP_PUSH_CONST ? bc[ip] % bb::P : bc[ip]));
                    ++ip;
                    break;
                case POWDR_OP_ADD: case POWDR_OP_SUB: case POWDR_OP_MUL: {
                    if (st.size() < 2) return false;
                    uint32_t r = st.back(); st.pop_back();
                    uint32_t l = st.back(); st.pop_back();
                    st.push_back(binary(op == POWDR_OP_ADD ? Node::ADD : op == POWDR_OP_SUB ? Node::SUB : Node::MUL, l, r));
                    break;
                }
                case POWDR_OP_NEG:
                case POWDR_OP_INV_OR_ZERO: {
                    if (st.empty()) return false;
                    uint32_t x = st.back(); st.pop_back();
                    st.push_back(unary(op == POWDR_OP_NEG ? Node::NEG : Node::INV, x));
                    break;
                }
                default: return false;
            }
            if (st.size() > (size_t)POWDR_EXPR_STACK_CAPACITY) return false;  // reference: stack overflow assert
        }
        if (st.size() != 1) return false;
        if (nodes_[st[0]].need > max_depth + 1) return false;
        gen(st[0], out, true);
        return true;
    }

private:
    std::vector<Node> nodes_;

    static uint32_t fadd(uint32_t a, uint32_t b) { return (uint32_t)(((uint64_t)a + b) % bb::P); }
    static uint32_t fsub(uint32_t a, uint32_t b) { return (uint32_t)(((uint64_t)a + bb::P - b) % bb::P); }
    static uint32_t fmul(uint32_t a, uint32_t b) { return (uint32_t)(((uint64_t)a * b) % bb::P); }
    static uint32_t finv(uint32_t a) {
        if (!a) return 0;
        uint32_t r = 1, e = bb::P - 2;
        while (e) { if (e & 1) r = fmul(r, a); a = fmul(a, a); e >>= 1; }
        return r;
    }
    bool is_const(uint32_t n, uint32_t v) const { return nodes_[n].kind == Node::CONST && nodes_[n].a == v; }
    bool is_leaf(uint32_t n) const { return nodes_[n].kind == Node::COL || nodes_[n].kind == Node::CONST; }
    uint32_t leaf(Node::Kind k, uint32_t a) { nodes_.push_back({k, a, 0, 1}); return (uint32_t)nodes_.size() - 1; }
    uint32_t unary(Node::Kind k, uint32_t x) {
        const Node& c = nodes_[x];
        if (c.kind == Node::CONST) return leaf(Node::CONST, k == Node::NEG ? fsub(0, c.a) : finv(c.a));
        if (k == Node::NEG && c.kind == Node::NEG) return c.a;  // --x
        nodes_.push_back({k, x, 0, nodes_[x].need});
        return (uint32_t)nodes_.size() - 1;
    }
    uint32_t binary(Node::Kind k, uint32_t l, uint32_t r) {
        const Node& L = nodes_[l];
        const Node& R = nodes_[r];
        if (L.kind == Node::CONST && R.kind == Node::CONST)
            return leaf(Node::CONST, k == Node::ADD ? fadd(L.a, R.a) : k == Node::SUB ? fsub(L.a, R.a) : fmul(L.a, R.a));
        if (k == Node::ADD) { if (is_const(l, 0)) return r; if (is_const(r, 0)) return l; }
        if (k == Node::SUB) { if (is_const(r, 0)) return l; if (is_const(l, 0)) return unary(Node::NEG, r); }
        if (k == Node::MUL) {
            if (is_const(l, 1)) return r;
            if (is_const(r, 1)) return l;
            if (is_const(l, 0) || is_const(r, 0)) return leaf(Node::CONST, 0);
        }
        uint8_t nl = nodes_[l].need, nr = nodes_[r].need, need;
        if (is_leaf(r)) need = nl;
        else if (is_leaf(l)) need = nr;
        else need = nl == nr ? (uint8_t)(nl + 1) : (nl > nr ? nl : nr);
        nodes_.push_back({k, l, r, need});
        return (uint32_t)nodes_.size() - 1;
    }

    static void emit(std::vector<uint32_t>& out, uint32_t op, uint32_t a) { out.push_back(op); out.push_back(a); }
    void emit_leaf(const Node& n, std::vector<uint32_t>& out, bool first) {
        if (n.kind == Node::COL) emit(out, first ? SET_COL : PUSH_COL, n.a);
        else emit(out, first ? SET_CONST : PUSH_CONST, bb::to_monty(n.a));
    }
    // `first`: the value stack is empty when this subtree starts (its first leaf uses SET instead of PUSH).
    // Explicit recursion is fine here: depth is bounded by the tree height of expressions the reference
    // already walked recursively (emit_expr), and the trees are rebuilt from <= 2^32-word programs.
    void gen(uint32_t n, std::vector<uint32_t>& out, bool first) {
        // iterative post-order with an explicit stack to survive very deep left-leaning chains
        struct Frame { uint32_t n; bool first; int state; };
        std::vector<Frame> st;
        st.push_back({n, first, 0});
        while (!st.empty()) {
            Frame f = st.back();
            st.pop_back();
            const Node nd = nodes_[f.n];
            if (nd.kind == Node::COL || nd.kind == Node::CONST) { emit_leaf(nd, out, f.first); continue; }
            if (nd.kind == Node::NEG || nd.kind == Node::INV) {
                if (f.state == 0) { st.push_back({f.n, f.first, 1}); st.push_back({nd.a, f.first, 0}); }
                else emit(out, nd.kind == Node::NEG ? NEG : INV, 0);
                continue;
            }
            const uint32_t l = nd.a, r = nd.b;
            const bool lr_leaf = is_leaf(r), ll_leaf = is_leaf(l);
            if (f.state == 0) {
                st.push_back({f.n, f.first, 1});
                if (lr_leaf) st.push_back({l, f.first, 0});
                else if (ll_leaf) st.push_back({r, f.first, 0});
                else if (nodes_[r].need > nodes_[l].need) { st.push_back({l, false, 0}); st.push_back({r, f.first, 0}); }  // r first
                else { st.push_back({r, false, 0}); st.push_back({l, f.first, 0}); }                                   // l first
                continue;
            }
            // operands are evaluated: combine
            if (lr_leaf) {
                const Node& R = nodes_[r];
                if (R.kind == Node::COL) emit(out, nd.kind == Node::ADD ? ADD_COL : nd.kind == Node::SUB ? SUB_COL : MUL_COL, R.a);
                else if (nd.kind == Node::ADD) emit(out, ADD_CONST, bb::to_monty(R.a));
                else if (nd.kind == Node::SUB) emit(out, ADD_CONST, bb::to_monty(fsub(0, R.a)));
                else emit(out, MUL_CONST, bb::to_monty(R.a));
            } else if (ll_leaf) {
                const Node& L = nodes_[l];
                if (L.kind == Node::COL) emit(out, nd.kind == Node::ADD ? ADD_COL : nd.kind == Node::SUB ? RSUB_COL : MUL_COL, L.a);
                else if (nd.kind == Node::ADD) emit(out, ADD_CONST, bb::to_monty(L.a));
                else if (nd.kind == Node::SUB) emit(out, RSUB_CONST, bb::to_monty(L.a));
                else emit(out, MUL_CONST, bb::to_monty(L.a));
            } else if (nodes_[r].need > nodes_[l].need) {
                // stack: [r, l(top)]
                emit(out, nd.kind == Node::ADD ? ADD : nd.kind == Node::SUB ? RSUB : MUL, 0);
            } else {
                // stack: [l, r(top)]
                emit(out, nd.kind == Node::ADD ? ADD : nd.kind == Node::SUB ? SUB : MUL, 0);
            }
        }
    }
};

}  // namespace xbc
