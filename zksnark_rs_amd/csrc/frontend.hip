// frontend.hip -- the input side of the hot path: the .zk language front end (host, as in the
// reference) and QAP::from(root_rep) for ARBITRARY roots on the GPU.
//
//   tokenizer / AST / ASTParser::try_parse  /root/reference/src/groth16/circuit/ast.rs:1-375,
//                                           /root/reference/src/groth16/circuit/mod.rs:224-527
//   circuit::weights                        /root/reference/src/groth16/circuit/mod.rs:529-656
//   QAP<CoefficientPoly<FrLocal>>::from     /root/reference/src/groth16/fr.rs:140-173
//   From<(roots,points)>, lagrange_basis    /root/reference/src/groth16/coefficient_poly.rs:159-190
//   root_poly                               /root/reference/src/groth16/coefficient_poly.rs:192-200
//
// The reference interpolates every non-zero entry separately (O(nnz n^2) field operations and
// n-1 inversions per entry).  Here t(X) = prod (X - r_j) is built once, each Lagrange basis
// polynomial is one synthetic division t / (X - r_k) scaled by 1 / t'(r_k) (one lane per root), and
// each wire polynomial is the weighted sum of the bases of its entries -- the same unique
// interpolating polynomials, so the dense coefficient matrices are identical.
#include <map>
#include <sstream>
#include <unordered_map>
#include "pipeline.hpp"

namespace zk {

// ---- host-side field helpers (same ff.cuh code, compiled for the host) -----------------------
static bool fr_parse_decimal(const std::string& s, Fr& out) {
    // bn's Fr::from_str: decimal digits only, value accumulated in the field (wraps mod r)
    Fr ten = host_fr_from_u64(10), acc = Fr::zero();
    for (char ch : s) {
        if (ch < '0' || ch > '9') return false;
        acc = acc * ten + host_fr_from_u64((uint64_t)(ch - '0'));
    }
    out = acc;
    return true;
}

struct ParseError {
    int status;
    std::string msg;
};

enum class Kw { In, Out, Verify, Program, Equal, Mul, Add };
struct Tok {
    enum Kind { Keyword, Var, Open, Close, Literal } kind;
    Kw kw{};
    std::string var;
    Fr lit{};
};
struct Node {
    enum Kind { In, Out, Verify, Program, Assign, Mul, Add, Var, Literal } kind;
    std::vector<Node> kids;
    std::string var;
    Fr lit{};
};

static void tokenize_word(std::string w, std::vector<Tok>& out, long line) {
    bool opened = false;
    if (!w.empty() && w[0] == '(') {
        Tok t; t.kind = Tok::Open; out.push_back(t);
        w = w.substr(1);
        opened = true;
    }
    auto syntax = [&](const char* m) { return ParseError{ZK_ERR_ARG, "SyntaxErr(" + std::to_string(line) + ", " + m + ")"}; };
    if (w.empty()) throw syntax("found whitespace after '('");
    static const std::map<std::string, Kw> kws = {{"in", Kw::In}, {"out", Kw::Out}, {"verify", Kw::Verify}, {"program", Kw::Program},
                                                  {"=", Kw::Equal}, {"*", Kw::Mul}, {"+", Kw::Add}};
    auto it = kws.find(w);
    if (it != kws.end()) { Tok t; t.kind = Tok::Keyword; t.kw = it->second; out.push_back(t); return; }
    if (w.find('(') != std::string::npos) throw syntax("unexpected '('");
    if (w.find_first_of("*+=") != std::string::npos) throw syntax("unexpected operator");
    size_t cut = w.find(')');
    std::string head = cut == std::string::npos ? w : w.substr(0, cut);
    std::string tail = cut == std::string::npos ? "" : w.substr(cut);
    if (opened && !tail.empty()) throw syntax("unexpected ')'");
    if (head.empty()) throw ParseError{ZK_ERR_ARG, "panic: empty token before ')'"};
    if (head[0] >= '0' && head[0] <= '9') {
        Tok t; t.kind = Tok::Literal;
        if (!fr_parse_decimal(head, t.lit)) throw syntax("could not parse literal");
        out.push_back(t);
    } else {
        Tok t; t.kind = Tok::Var; t.var = head; out.push_back(t);
    }
    for (char c : tail) {
        if (c != ')') throw syntax("expected ')'");
        Tok t; t.kind = Tok::Close; out.push_back(t);
    }
}

static std::vector<Tok> tokenize(const std::string& code) {
    std::vector<Tok> toks;
    long line = 1;
    std::istringstream all(code);
    std::string ln;
    while (std::getline(all, ln)) {
        std::istringstream ws(ln);
        std::string w;
        while (ws >> w) tokenize_word(w, toks, line);
        ++line;
    }
    return toks;
}

static std::vector<Tok> take_group(const std::vector<Tok>& t, size_t& pos) {
    std::vector<Tok> g;
    if (pos >= t.size()) return g;
    const Tok& first = t[pos++];
    if (first.kind == Tok::Open) {
        int depth = 1;
        while (pos < t.size()) {
            const Tok& u = t[pos++];
            if (u.kind == Tok::Open) ++depth;
            if (u.kind == Tok::Close && --depth == 0) break;
            g.push_back(u);
        }
        return g;
    }
    if (first.kind == Tok::Var || first.kind == Tok::Literal) { g.push_back(first); return g; }
    throw ParseError{ZK_ERR_ARG, "panic: Cannot parse malformed group"};
}

static Node parse_node(const std::vector<Tok>& t) {
    auto bad = [](const std::string& m) { return ParseError{ZK_ERR_ARG, "StructureErr(None, " + m + ")"}; };
    if (t.empty()) throw bad("Malformed expression");
    size_t pos = 1;
    const Tok& h = t[0];
    Node n;
    if (h.kind == Tok::Var) { n.kind = Node::Var; n.var = h.var; return n; }
    if (h.kind == Tok::Literal) { n.kind = Node::Literal; n.lit = h.lit; return n; }
    if (h.kind != Tok::Keyword) throw bad("Malformed expression");
    switch (h.kw) {
        case Kw::In: case Kw::Out: case Kw::Verify:
            n.kind = h.kw == Kw::In ? Node::In : h.kw == Kw::Out ? Node::Out : Node::Verify;
            for (; pos < t.size(); ++pos) {
                if (t[pos].kind != Tok::Var) throw bad("Non variable found in declaration");
                Node v; v.kind = Node::Var; v.var = t[pos].var; n.kids.push_back(v);
            }
            return n;
        case Kw::Program: case Kw::Add:
            n.kind = h.kw == Kw::Program ? Node::Program : Node::Add;
            for (;;) {
                auto g = take_group(t, pos);
                if (g.empty()) break;
                n.kids.push_back(parse_node(g));
            }
            return n;
        case Kw::Equal: {
            auto lhs = take_group(t, pos);
            if (lhs.size() != 1 || lhs[0].kind != Tok::Var) throw bad("Can only assign to a variable");
            Node l; l.kind = Node::Var; l.var = lhs[0].var;
            n.kind = Node::Assign;
            n.kids = {l, parse_node(take_group(t, pos))};
            return n;
        }
        case Kw::Mul: {
            Node a = parse_node(take_group(t, pos));
            Node b = parse_node(take_group(t, pos));
            n.kind = Node::Mul;
            n.kids = {a, b};
            return n;
        }
    }
    throw bad("Malformed expression");
}

static std::vector<Node> parse_top(const std::vector<Tok>& toks) {
    std::vector<Node> out;
    size_t pos = 0;
    for (;;) {
        auto g = take_group(toks, pos);
        if (g.empty()) break;
        out.push_back(parse_node(g));
    }
    return out;
}

}  // namespace zk

// DummyRep (dummy_rep.rs:6-13) for Fr + what circuit::weights needs
struct zk_circuit {
    typedef std::vector<std::pair<uint32_t, zk::Fr>> Row;   // (gate index 0-based, value in Montgomery form)
    std::vector<Row> u, v, w;
    size_t n_gates = 0, input = 0;
    std::vector<zk::Node> exprs;
    std::vector<std::string> order;    // variable_order (ast.rs:62-83)
    std::string last_error;
};

namespace zk {

static zk_circuit* circuit_parse(const std::string& code) {
    std::unique_ptr<zk_circuit> c(new zk_circuit());
    auto toks = tokenize(code);
    c->exprs = parse_top(toks);
    {   // variable order: first appearance after the `verify` keyword
        std::unordered_map<std::string, bool> seen;
        size_t i = 0;
        while (i < toks.size() && !(toks[i].kind == Tok::Keyword && toks[i].kw == Kw::Verify)) ++i;
        for (; i < toks.size(); ++i)
            if (toks[i].kind == Tok::Var && !seen.count(toks[i].var)) { seen[toks[i].var] = true; c->order.push_back(toks[i].var); }
    }
    size_t gate = 0;
    auto serr = [&](const std::string& m) { return ParseError{ZK_ERR_ARG, "StructureErr(" + std::to_string(gate) + ", " + m + ")"}; };
    auto& ex = c->exprs;
    if (ex.size() != 4) throw serr("Expected exactly one each of 'in', 'out', 'verify' and 'program'");
    if (ex[0].kind != Node::In) throw serr("Expected first expression to be 'in'");
    if (ex[1].kind != Node::Out) throw serr("Expected second expression to be 'out'");
    if (ex[2].kind != Node::Verify) throw serr("Expected third expression to be 'verify'");
    if (ex[3].kind != Node::Program) throw serr("Expected fourth expression to be 'program'");
    std::unordered_map<std::string, size_t> wire;
    c->u.resize(1); c->v.resize(1); c->w.resize(1);     // wire 0 = the constant 1
    for (const auto& var : ex[2].kids) {
        wire[var.var] = c->u.size();
        c->u.emplace_back(); c->v.emplace_back(); c->w.emplace_back();
        ++c->input;
    }
    const Fr one = Fr::one();
    auto add_wire = [&]() { c->u.emplace_back(); c->v.emplace_back(); c->w.emplace_back(); };
    auto touch = [&](std::vector<zk_circuit::Row>& side, const std::string& name, const Fr& coeff) {
        auto it = wire.find(name);
        if (it == wire.end()) {
            wire[name] = side.size();
            add_wire();
            side.back().push_back({(uint32_t)(gate - 1), coeff});
        } else {
            side[it->second].push_back({(uint32_t)(gate - 1), coeff});
        }
    };
    auto side_input = [&](std::vector<zk_circuit::Row>& side, const Node& e) {
        if (e.kind == Node::Literal) side[0].push_back({(uint32_t)(gate - 1), e.lit});
        else if (e.kind == Node::Var) touch(side, e.var, one);
        else if (e.kind == Node::Add) {
            for (const auto& t : e.kids) {
                if (t.kind == Node::Literal) side[0].push_back({(uint32_t)(gate - 1), t.lit});
                else if (t.kind == Node::Var) touch(side, t.var, one);
                else if (t.kind == Node::Mul) {
                    if (t.kids[0].kind != Node::Literal) throw serr("LHS of a '*' expression in a '+' expression must be a literal");
                    if (t.kids[1].kind != Node::Var) throw serr("RHS of a '*' expression in a '+' expression must be a variable");
                    touch(side, t.kids[1].var, t.kids[0].lit);
                } else throw serr("Invalid expression found in '+' expression");
            }
        } else throw serr("Invalid expression found in '*' expression");
    };
    for (const auto& asg : ex[3].kids) {
        ++gate;
        if (asg.kind != Node::Assign) throw serr("Program expression must be a list of '=' expressions");
        const std::string& out = asg.kids[0].var;
        auto it = wire.find(out);
        if (it == wire.end()) {
            wire[out] = c->u.size();
            add_wire();
            c->w.back().push_back({(uint32_t)(gate - 1), one});
        } else if (it->second <= c->input) {
            if (!c->w[it->second].empty()) throw serr("Varify variable cannot be the output of two different gates");
            c->w[it->second].push_back({(uint32_t)(gate - 1), one});
        } else {
            throw serr("Already declared variable cannot be the output wire of a gate");
        }
        const Node& rhs = asg.kids[1];
        if (rhs.kind == Node::Mul) {
            side_input(c->u, rhs.kids[0]);
            side_input(c->v, rhs.kids[1]);
        }
    }
    c->n_gates = gate;
    return c.release();
}

static bool eval_node(const Node& e, const std::unordered_map<std::string, Fr>& env, Fr& out) {
    switch (e.kind) {
        case Node::Literal: out = e.lit; return true;
        case Node::Var: { auto it = env.find(e.var); if (it == env.end()) return false; out = it->second; return true; }
        case Node::Mul: {
            Fr a, b;
            if (!eval_node(e.kids[0], env, a) || !eval_node(e.kids[1], env, b)) return false;
            out = a * b;
            return true;
        }
        case Node::Add: {
            Fr acc = Fr::zero();
            for (const auto& k : e.kids) { Fr t; if (!eval_node(k, env, t)) return false; acc = acc + t; }
            out = acc;
            return true;
        }
        default: return false;
    }
}

// circuit::weights (circuit/mod.rs:529-637): inputs in `in` order -> [1] ++ values in variable order
static void circuit_weights(const zk_circuit& c, const uint64_t* inputs, size_t n_in, uint64_t* out, size_t m) {
    auto serr = [](const std::string& s) { return ParseError{ZK_ERR_ARG, "StructureErr(None, " + s + ")"}; };
    const auto& ex = c.exprs;
    if (ex[0].kids.size() != n_in) throw serr("Wrong number of values supplied");
    if (m != c.u.size()) throw serr("weights buffer size mismatch");
    std::unordered_map<std::string, Fr> env;
    for (size_t i = 0; i < n_in; ++i) {
        Fr x;
        for (int k = 0; k < 4; ++k) { x.l[2 * k] = (uint32_t)inputs[4 * i + k]; x.l[2 * k + 1] = (uint32_t)(inputs[4 * i + k] >> 32); }
        if (!x.raw_in_range()) throw ParseError{ZK_ERR_RANGE, "input value >= r"};
        env[ex[0].kids[i].var] = Fr::from_canonical(x);
    }
    for (const auto& a : ex[3].kids) {
        if (a.kind != Node::Assign) throw serr("Program expression must be a list of '=' expressions");
        const std::string& var = a.kids[0].var;
        if (env.count(var)) throw serr("Attempted to assign to an already assigned variable");
        Fr val;
        if (!eval_node(a.kids[1], env, val)) throw serr("Under constrained expression");
        env[var] = val;
    }
    auto put = [&](size_t i, const Fr& mont) {
        Fr x = mont.to_canonical();
        for (int k = 0; k < 4; ++k) out[4 * i + k] = (uint64_t)x.l[2 * k] | ((uint64_t)x.l[2 * k + 1] << 32);
    };
    put(0, Fr::one());
    if (c.order.size() + 1 != m) throw ParseError{ZK_ERR_ARG, "panic: variable order does not cover every wire"};
    for (size_t i = 0; i < c.order.size(); ++i) {
        auto it = env.find(c.order[i]);
        if (it == env.end()) throw ParseError{ZK_ERR_ARG, "panic: Every variable should have an assignment"};
        put(i + 1, it->second);
    }
}

// ---- QAP::from(root_rep) for arbitrary roots on the GPU ----------------------------------------
// t(X) = prod_j (X - r_j): one workgroup, n rounds of a parallel "multiply by (X - r_j)"
__global__ __launch_bounds__(1024) void k_root_poly(const Fr* __restrict__ roots, size_t n, Fr* __restrict__ t /* n+1 */, Fr* __restrict__ tmp) {
    for (size_t k = threadIdx.x; k <= n; k += blockDim.x) t[k] = k == 0 ? Fr::one() : Fr::zero();
    __syncthreads();
    for (size_t j = 0; j < n; ++j) {
        Fr r = roots[j];
        // new[k] = old[k-1] - r * old[k],  degree grows from j to j+1
        for (size_t k = threadIdx.x; k <= j + 1; k += blockDim.x) {
            Fr lo = k <= j ? t[k] : Fr::zero();
            Fr hi = k >= 1 ? t[k - 1] : Fr::zero();
            tmp[k] = hi - r * lo;
        }
        __syncthreads();
        for (size_t k = threadIdx.x; k <= j + 1; k += blockDim.x) t[k] = tmp[k];
        __syncthreads();
    }
}
// basis[k][0..n) = coefficients of L_k(X) = t(X) / ((X - r_k) t'(r_k)); one lane per root
__global__ void k_lagrange_bases(const Fr* __restrict__ roots, const Fr* __restrict__ t, size_t n, Fr* __restrict__ basis, int* __restrict__ flag) {
    size_t k = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    Fr r = roots[k];
    Fr* q = basis + k * n;
    // synthetic division: q[n-1] = t[n]; q[i-1] = t[i] + r q[i]
    Fr carry = t[n];
    for (size_t i = n; i-- > 0;) {
        q[i] = carry;
        carry = t[i] + r * carry;
    }
    // t'(r_k) = q(r_k)
    Fr d = Fr::zero();
    for (size_t i = n; i-- > 0;) d = d * r + q[i];
    if (d.is_zero()) { atomicOr(flag, 1); return; }   // repeated root
    Fr dinv = d.inv();
    for (size_t i = 0; i < n; ++i) q[i] = q[i] * dinv;
}
// dense[i][k] = sum_e val_e * basis[gate_e][k]
__global__ void k_rows_to_dense(const uint32_t* __restrict__ ptr, const uint32_t* __restrict__ gate, const Fr* __restrict__ val,
                                const Fr* __restrict__ basis, size_t m, size_t n, Fr* __restrict__ dense) {
    size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= m * n) return;
    size_t i = idx / n, k = idx % n;
    Fr acc = Fr::zero();
    for (uint32_t e = ptr[i]; e < ptr[i + 1]; ++e) acc = acc + val[e] * basis[(size_t)gate[e] * n + k];
    dense[idx] = acc;
}

static zk_qap* circuit_to_qap(zk_ctx* ctx, const zk_circuit& c) {
    const size_t n = c.n_gates, m = c.u.size();
    ZK_REQUIRE(n >= 1, ZK_ERR_ARG, "circuit has no gates");
    ZK_REQUIRE(n <= 16384, ZK_ERR_SIZE, "dense QAP construction supports at most 16384 gates (3 m n field elements); use the sparse roots-of-unity form");
    hipStream_t st = ctx->stream;
    std::unique_ptr<zk_qap> q(new zk_qap());
    q->ctx = ctx; q->dense = true; q->n = n; q->m = m; q->input = c.input;
    // roots 1..n (circuit/mod.rs:517)
    std::vector<Fr> roots(n);
    for (size_t j = 0; j < n; ++j) roots[j] = host_fr_from_u64(j + 1);
    DevBuf<Fr> d_roots(n), d_tmp(n + 1), d_basis(n * n);
    DevBuf<int> flag(1);
    ZK_HIP(hipMemsetAsync(flag.p, 0, sizeof(int), st));
    ZK_HIP(hipMemcpyAsync(d_roots.p, roots.data(), n * sizeof(Fr), hipMemcpyHostToDevice, st));
    q->dt.alloc(n + 1);
    hipLaunchKernelGGL(k_root_poly, dim3(1), dim3(1024), 0, st, d_roots.p, n, q->dt.p, d_tmp.p);
    hipLaunchKernelGGL(k_lagrange_bases, dim3(ceil_div(n, 64)), dim3(64), 0, st, d_roots.p, q->dt.p, n, d_basis.p, flag.p);
    ZK_HIP(hipGetLastError());
    auto build = [&](const std::vector<zk_circuit::Row>& rows, DevBuf<Fr>& dense) {
        std::vector<uint32_t> ptr(m + 1, 0), gate;
        std::vector<Fr> val;
        for (size_t i = 0; i < m; ++i) {
            for (const auto& e : rows[i]) { gate.push_back(e.first); val.push_back(e.second); }
            ptr[i + 1] = (uint32_t)gate.size();
        }
        DevBuf<uint32_t> d_ptr(m + 1), d_gate(std::max<size_t>(gate.size(), 1));
        DevBuf<Fr> d_val(std::max<size_t>(val.size(), 1));
        ZK_HIP(hipMemcpyAsync(d_ptr.p, ptr.data(), ptr.size() * 4, hipMemcpyHostToDevice, st));
        if (!gate.empty()) {
            ZK_HIP(hipMemcpyAsync(d_gate.p, gate.data(), gate.size() * 4, hipMemcpyHostToDevice, st));
            ZK_HIP(hipMemcpyAsync(d_val.p, val.data(), val.size() * sizeof(Fr), hipMemcpyHostToDevice, st));
        }
        dense.alloc(m * n);
        hipLaunchKernelGGL(k_rows_to_dense, dim3(ceil_div(m * n, 256)), dim3(256), 0, st, d_ptr.p, d_gate.p, d_val.p, d_basis.p, m, n, dense.p);
        ZK_HIP(hipGetLastError());
        ZK_HIP(hipStreamSynchronize(st));
    };
    build(c.u, q->du);
    build(c.v, q->dv);
    build(c.w, q->dw);
    int h = 0;
    ZK_HIP(hipMemcpy(&h, flag.p, sizeof(int), hipMemcpyDeviceToHost));
    ZK_REQUIRE(!h, ZK_ERR_DIV_BY_ZERO, "repeated root in the QAP domain");
    q->t_degree = n;          // t is monic of degree n
    q->t_is_zero = false;
    q->t_cinv.alloc(1);
    Fr one = Fr::one();
    ZK_HIP(hipMemcpy(q->t_cinv.p, &one, sizeof(Fr), hipMemcpyHostToDevice));
    return q.release();
}

// The same QAP without the interpolation: the root representation's rows go to the device as they are and the prover works on
// the integer roots 1..n in the evaluation basis (aproots.hip).  No 16384-gate limit; proofs are byte-identical to the dense form's.
static zk_qap* circuit_to_qap_sparse(zk_ctx* ctx, const zk_circuit& c) {
    const size_t n = c.n_gates, m = c.u.size();
    ZK_REQUIRE(n >= 1, ZK_ERR_ARG, "circuit has no gates");
    struct HostRows { std::vector<uint64_t> ptr, val; std::vector<uint32_t> gate; };
    HostRows h[3];
    const std::vector<zk_circuit::Row>* src[3] = {&c.u, &c.v, &c.w};
    for (int k = 0; k < 3; ++k) {
        h[k].ptr.assign(m + 1, 0);
        for (size_t i = 0; i < m; ++i) {
            for (const auto& e : (*src[k])[i]) {
                h[k].gate.push_back(e.first);
                const Fr v = e.second.to_canonical();
                for (int j = 0; j < 4; ++j) h[k].val.push_back((uint64_t)v.l[2 * j] | ((uint64_t)v.l[2 * j + 1] << 32));
            }
            h[k].ptr[i + 1] = h[k].gate.size();
        }
        if (h[k].gate.empty()) { h[k].gate.push_back(0); h[k].val.assign(4, 0); }   // never a null array
    }
    zk_qap_sparse_desc d{};
    d.m = m; d.input = c.input;
    d.u = {h[0].ptr.data(), h[0].gate.data(), h[0].val.data()};
    d.v = {h[1].ptr.data(), h[1].gate.data(), h[1].val.data()};
    d.w = {h[2].ptr.data(), h[2].gate.data(), h[2].val.data()};
    return qap_upload_sparse_integers(ctx, d, n);
}

static void qap_download_dense(zk_ctx* ctx, const zk_qap& q, uint64_t* u, uint64_t* v, uint64_t* w, uint64_t* t) {
    ZK_REQUIRE(q.dense, ZK_ERR_ARG, "zk_qap_download_dense: QAP is in sparse form");
    auto down = [&](const DevBuf<Fr>& src, uint64_t* dst, size_t count) {
        if (!dst) return;
        DevBuf<Fr> tmp(count);
        fr_from_mont(ctx, src.p, tmp.p, count);
        ZK_HIP(hipMemcpyAsync(dst, tmp.p, count * sizeof(Fr), hipMemcpyDeviceToHost, ctx->stream));
        ZK_HIP(hipStreamSynchronize(ctx->stream));
    };
    down(q.du, u, q.m * q.n);
    down(q.dv, v, q.m * q.n);
    down(q.dw, w, q.m * q.n);
    down(q.dt, t, q.n + 1);
}

template <class Fn>
static int front_guard(std::string* err, Fn&& fn) {
    try { fn(); return ZK_OK; }
    catch (const ParseError& e) { if (err) *err = e.msg; return e.status; }
    catch (const std::exception& e) { if (err) *err = e.what(); return ZK_ERR_ARG; }
    catch (...) { if (err) *err = "unknown error"; return ZK_ERR_ARG; }
}

}  // namespace zk

using namespace zk;

extern "C" {

int zk_circuit_parse(const char* code, zk_circuit** out, char* err, size_t err_len) {
    if (!code || !out) return ZK_ERR_ARG;
    *out = nullptr;
    std::string msg;
    int rc = front_guard(&msg, [&] { *out = circuit_parse(code); });
    if (err && err_len) { std::snprintf(err, err_len, "%s", msg.c_str()); }
    return rc;
}
void zk_circuit_free(zk_circuit* c) { delete c; }
int zk_circuit_dims(const zk_circuit* c, size_t* m, size_t* n, size_t* input, size_t* n_in) {
    if (!c) return ZK_ERR_ARG;
    if (m) *m = c->u.size();
    if (n) *n = c->n_gates;
    if (input) *input = c->input;
    if (n_in) *n_in = c->exprs[0].kids.size();
    return ZK_OK;
}
// which: 0 = u, 1 = v, 2 = w.  ptr has m+1 entries; gate/val receive nnz entries (pass NULL to query nnz).
int zk_circuit_rows(const zk_circuit* c, int which, uint64_t* ptr, uint32_t* gate, uint64_t* val, size_t* nnz) {
    if (!c || which < 0 || which > 2) return ZK_ERR_ARG;
    const auto& rows = which == 0 ? c->u : which == 1 ? c->v : c->w;
    size_t count = 0;
    for (size_t i = 0; i < rows.size(); ++i) {
        if (ptr) ptr[i] = count;
        for (const auto& e : rows[i]) {
            if (gate) gate[count] = e.first;
            if (val) {
                Fr x = e.second.to_canonical();
                for (int k = 0; k < 4; ++k) val[4 * count + k] = (uint64_t)x.l[2 * k] | ((uint64_t)x.l[2 * k + 1] << 32);
            }
            ++count;
        }
    }
    if (ptr) ptr[rows.size()] = count;
    if (nnz) *nnz = count;
    return ZK_OK;
}
int zk_circuit_weights(const zk_circuit* c, const uint64_t* inputs, size_t n_in, uint64_t* weights_out, size_t m) {
    if (!c || !weights_out || (n_in && !inputs)) return ZK_ERR_ARG;
    std::string msg;
    int rc = front_guard(&msg, [&] { circuit_weights(*c, inputs, n_in, weights_out, m); });
    const_cast<zk_circuit*>(c)->last_error = msg;
    return rc;
}
const char* zk_circuit_last_error(const zk_circuit* c) { return c ? c->last_error.c_str() : "null circuit"; }

int zk_circuit_qap(zk_ctx* ctx, const zk_circuit* c, zk_qap** out) {
    if (!ctx || !c || !out) return ZK_ERR_ARG;
    *out = nullptr;
    return guarded(ctx, [&] { *out = circuit_to_qap(ctx, *c); });
}
int zk_circuit_qap_sparse(zk_ctx* ctx, const zk_circuit* c, zk_qap** out) {
    if (!ctx || !c || !out) return ZK_ERR_ARG;
    *out = nullptr;
    return guarded(ctx, [&] { *out = circuit_to_qap_sparse(ctx, *c); });
}
int zk_qap_download_dense(zk_ctx* ctx, const zk_qap* qap, uint64_t* u, uint64_t* v, uint64_t* w, uint64_t* t) {
    if (!ctx || !qap) return ZK_ERR_ARG;
    return guarded(ctx, [&] { qap_download_dense(ctx, *qap, u, v, w, t); });
}
int zk_qap_dims(const zk_qap* qap, size_t* n, size_t* m, size_t* input, int* dense) {
    if (!qap) return ZK_ERR_ARG;
    if (n) *n = qap->n;
    if (m) *m = qap->m;
    if (input) *input = qap->input;
    if (dense) *dense = qap->dense ? 1 : 0;
    return ZK_OK;
}

int zk_qap_weighted_sum(zk_ctx* ctx, const zk_qap* qap, const uint64_t* weights, size_t m, int which, uint64_t* out) {
    if (!ctx || !qap) return ZK_ERR_ARG;
    return guarded(ctx, [&] { qap_weighted_sum(ctx, *qap, weights, m, which, out); });
}

/* 0: sparse rows over the roots of unity w^j (n = 2^k), 1: dense coefficient matrices, 2: sparse rows over the integers 1..n, 3: over the caller's roots */
int zk_qap_kind(const zk_qap* qap) {
    if (!qap) return ZK_ERR_ARG;
    return qap->dense ? 1 : (qap->roots == 2 ? 3 : qap->roots ? 2 : 0);
}

}  // extern "C"
