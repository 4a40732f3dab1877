// reference_tests.cpp -- the reference crate's own BN254 end-to-end tests, restated against the C++ host
// API (include/zksnark.hpp) that mirrors groth16::{setup, prove, verify}, ASTParser, QAP and FrLocal.
// Built with g++ and linked against libzkgpu.so by tests/test_cpp_api.py (-m gpu); prints one
// "ok <name>" line per test and exits non-zero on the first failed assertion.
//
//   simple_circuit_test        /root/reference/src/lib.rs:156-190
//   single_mult_honest_bn      /root/reference/src/groth16/fr.rs:248-270
//   bn_encrypt_quad_test       /root/reference/src/groth16/fr.rs:272-302
//   bn_encrypt_cubic_test      /root/reference/src/groth16/fr.rs:304-358
//   bn_encrypt_deg_15_test     /root/reference/src/groth16/fr.rs:360-416
//   exp_encrypted / field laws are covered through the C ABI in tests/test_gpu_blocks.py
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <sstream>
#include <string>

#include "zksnark.hpp"

using namespace zksnark;
using groth16::prove;
using groth16::setup;
using groth16::verify;

static std::string programs_dir;
static std::string read_to_string(const std::string& name) {
    std::ifstream f(programs_dir + "/" + name);
    if (!f) { std::fprintf(stderr, "cannot read %s/%s\n", programs_dir.c_str(), name.c_str()); std::exit(2); }
    std::stringstream ss;
    ss << f.rdbuf();
    return ss.str();
}
#define ASSERT(cond)                                                                    \
    do {                                                                                \
        if (!(cond)) { std::fprintf(stderr, "assertion failed at %s:%d: %s\n", __FILE__, __LINE__, #cond); std::exit(1); } \
    } while (0)

static void simple_circuit_test(const Context& ctx) {
    // x = 4ab + c + 6
    std::string code = read_to_string("simple.zk");
    QAP qap = QAP::from(ctx, ASTParser::try_parse(code));
    // The assignments are the inputs to the circuit in the order they appear in the file
    std::vector<FrLocal> assignments = {3, 2, 4};   // a, b, c
    auto weights = groth16::weights(code, assignments);
    ASSERT((weights == std::vector<FrLocal>{1, 2, 34, 6, 3, 4}));   // circuit/mod.rs:745-769

    auto sigma = setup(ctx, qap);
    auto proof = prove(ctx, qap, sigma, weights);
    ASSERT(verify(ctx, sigma, {FrLocal(2), FrLocal(34)}, proof));

    auto sigma2 = setup(ctx, qap);
    auto proof2 = prove(ctx, qap, sigma2, weights);
    ASSERT(!verify(ctx, sigma2, {FrLocal(2), FrLocal(25)}, proof2));
    std::puts("ok simple_circuit_test");
}

static void single_mult_honest_bn(const Context& ctx) {
    auto constant = [](uint64_t v) { return std::vector<FrLocal>{FrLocal(v)}; };
    QAP qap = QAP::from_dense(ctx,
                              {constant(0), constant(0), constant(1), constant(0)},    // u
                              {constant(0), constant(0), constant(0), constant(1)},    // v
                              {constant(0), constant(1), constant(0), constant(0)},    // w
                              {FrLocal(250), FrLocal(1)},                              // t
                              2);                                                      // input; degree = 1
    std::vector<FrLocal> weights = {1, 51, 3, 17};
    for (int k = 0; k < 10; ++k) {
        auto sigma = setup(ctx, qap);
        auto proof = prove(ctx, qap, sigma, weights);
        ASSERT(verify(ctx, sigma, {FrLocal(51), FrLocal(3)}, proof));
    }
    std::puts("ok single_mult_honest_bn");
}

static void bn_encrypt_quad_test(const Context& ctx) {
    QAP qap = QAP::from(ctx, ASTParser::try_parse(read_to_string("lispesque_quad.zk")));
    Field F(ctx);
    for (int k = 0; k < 10; ++k) {
        FrLocal x = FrLocal::random_elem(), a = FrLocal::random_elem(), b = FrLocal::random_elem(), c = FrLocal::random_elem();
        FrLocal ax = F.mul(a, x), axb = F.add(ax, b);
        FrLocal share = F.add(F.add(F.mul(ax, x), F.mul(b, x)), c);            // a x^2 + b x + c
        // The order of the weights is determined by the order that the variables appear in the file
        std::vector<FrLocal> weights = {1, x, share, ax, a, F.mul(x, axb), b, c};
        auto sigma = setup(ctx, qap);
        auto proof = prove(ctx, qap, sigma, weights);
        ASSERT(verify(ctx, sigma, {x, share}, proof));
    }
    std::puts("ok bn_encrypt_quad_test");
}

static void bn_encrypt_cubic_test(const Context& ctx) {
    QAP qap = QAP::from(ctx, ASTParser::try_parse(read_to_string("lispesque_cubic.zk")));
    Field F(ctx);
    for (int k = 0; k < 10; ++k) {
        FrLocal x = FrLocal::random_elem(), a = FrLocal::random_elem(), b = FrLocal::random_elem(), c = FrLocal::random_elem(),
                d = FrLocal::random_elem();
        FrLocal ax = F.mul(a, x), t2 = F.mul(x, F.add(ax, b)), t3 = F.mul(x, F.add(t2, c));
        FrLocal share = F.add(t3, d);                                           // a x^3 + b x^2 + c x + d
        std::vector<FrLocal> weights = {1, x, share, ax, a, t2, b, t3, c, d};
        auto sigma = setup(ctx, qap);
        auto proof = prove(ctx, qap, sigma, weights);
        ASSERT(verify(ctx, sigma, {x, share}, proof));
        // a wrong public value must not verify
        ASSERT(!verify(ctx, sigma, {x, F.add(share, FrLocal(1))}, proof));
    }
    std::puts("ok bn_encrypt_cubic_test");
}

static void bn_encrypt_deg_15_test(const Context& ctx) {
    std::string code = read_to_string("deg_15.zk");
    QAP qap = QAP::from(ctx, ASTParser::try_parse(code));
    for (int k = 0; k < 10; ++k) {
        std::vector<FrLocal> inputs(17);
        for (auto& v : inputs) v = FrLocal::random_elem();
        auto weights = groth16::weights(code, inputs);
        auto sigma = setup(ctx, qap);
        auto proof = prove(ctx, qap, sigma, weights);
        ASSERT(verify(ctx, sigma, {weights[1], weights[2]}, proof));
    }
    std::puts("ok bn_encrypt_deg_15_test");
}

// single_mult_honest_bn's QAP written as the root representation it comes from (one gate at the root -250, circuit/mod.rs:201-214) and,
// on a second circuit, roots that are neither 1..n nor roots of unity: QAP::from_root_rep (zk_qap_upload_sparse_roots) gives the bytes of
// the dense struct literal under the same CRS and (r, s), and its own setup verifies.
static void root_representation_over_any_roots(const Context& ctx) {
    auto constant = [](uint64_t v) { return std::vector<FrLocal>{FrLocal(v)}; };
    QAP dense = QAP::from_dense(ctx, {constant(0), constant(0), constant(1), constant(0)}, {constant(0), constant(0), constant(0), constant(1)},
                                {constant(0), constant(1), constant(0), constant(0)}, {FrLocal(250), FrLocal(1)}, 2);
    FrLocal root;
    root.w = FrLocal::MODULUS;
    root.w[0] -= 250;
    QAP::Rows u(4), v(4), w(4);
    u[2] = {{root, FrLocal(1)}}; v[3] = {{root, FrLocal(1)}}; w[1] = {{root, FrLocal(1)}};
    QAP sparse = QAP::from_root_rep(ctx, {root}, u, v, w, 2);
    std::vector<FrLocal> weights = {1, 51, 3, 17};
    auto sigma = setup(ctx, dense);
    FrLocal r = FrLocal::random_elem(), s = FrLocal::random_elem();
    ASSERT(groth16::prove_with(ctx, dense, sigma, weights, r, s) == groth16::prove_with(ctx, sparse, sigma, weights, r, s));
    auto sigma2 = setup(ctx, sparse);
    ASSERT(verify(ctx, sigma2, {FrLocal(51), FrLocal(3)}, prove(ctx, sparse, sigma2, weights)));
    ASSERT(!verify(ctx, sigma2, {FrLocal(52), FrLocal(3)}, prove(ctx, sparse, sigma2, weights)));
    // two gates at the roots 7 and 1000003: out = (a * b) * c;  wires 1, out, a, b, c, t
    std::vector<FrLocal> roots = {FrLocal(7), FrLocal(1000003)};
    QAP::Rows u2(6), v2(6), w2(6);
    u2[2] = {{roots[0], FrLocal(1)}}; v2[3] = {{roots[0], FrLocal(1)}}; w2[5] = {{roots[0], FrLocal(1)}};   // a * b = t
    u2[5] = {{roots[1], FrLocal(1)}}; v2[4] = {{roots[1], FrLocal(1)}}; w2[1] = {{roots[1], FrLocal(1)}};   // t * c = out
    QAP q2 = QAP::from_root_rep(ctx, roots, u2, v2, w2, 1);
    auto s3 = setup(ctx, q2);
    std::vector<FrLocal> wt = {1, 3 * 5 * 11, 3, 5, 11, 15};
    ASSERT(verify(ctx, s3, {FrLocal(165)}, prove(ctx, q2, s3, wt)));
    ASSERT(!verify(ctx, s3, {FrLocal(166)}, prove(ctx, q2, s3, wt)));
    std::puts("ok root_representation_over_any_roots");
}

static void error_behaviour(const Context& ctx) {
    Field F(ctx);
    bool threw = false;
    try { F.div(FrLocal(1), FrLocal(0)); } catch (const Error& e) { threw = e.status == ZK_ERR_DIV_BY_ZERO; }   // fr.rs:54,69 panic
    ASSERT(threw);
    threw = false;
    try { ASTParser::try_parse("(in a b) (out x) (verify x) (program (= x (* a"); } catch (const Error& e) { threw = true; }   // ParseErr
    ASSERT(threw);
    // a CRS written to disk proves and verifies like the original (no analogue in the reference)
    std::string code = read_to_string("simple.zk");
    QAP qap = QAP::from(ctx, ASTParser::try_parse(code));
    auto weights = groth16::weights(code, {3, 2, 4});
    auto sigma = setup(ctx, qap);
    const char* tmp = std::getenv("ZK_TEST_TMP");
    std::string path = std::string(tmp ? tmp : "/tmp") + "/simple.zkcrs";
    sigma.save(ctx, path);
    auto sigma2 = groth16::Sigma::load(ctx, path);
    FrLocal r = FrLocal::random_elem(), s = FrLocal::random_elem();
    ASSERT(groth16::prove_with(ctx, qap, sigma, weights, r, s) == groth16::prove_with(ctx, qap, sigma2, weights, r, s));
    ASSERT(verify(ctx, sigma2, {FrLocal(2), FrLocal(34)}, prove(ctx, qap, sigma, weights)));
    std::puts("ok error_behaviour");
}

int main(int argc, char** argv) {
    programs_dir = argc > 1 ? argv[1] : "tests/golden/zk";
    try {
        Context ctx(0);
        simple_circuit_test(ctx);
        single_mult_honest_bn(ctx);
        bn_encrypt_quad_test(ctx);
        bn_encrypt_cubic_test(ctx);
        bn_encrypt_deg_15_test(ctx);
        root_representation_over_any_roots(ctx);
        error_behaviour(ctx);
    } catch (const Error& e) {
        std::fprintf(stderr, "zksnark::Error %d: %s\n", e.status, e.what());
        return 3;
    }
    std::puts("all ok");
    return 0;
}
