// oracle/zkparse.hpp -- CPU ORACLE (TEST INFRASTRUCTURE, NOT PRODUCT CODE).
//
// Restatement of the reference's .zk front end (needed to load BASELINE configs 1-2):
//   tokenizer / AST   /root/reference/src/groth16/circuit/ast.rs:1-375
//   ASTParser::try_parse -> DummyRep   /root/reference/src/groth16/circuit/mod.rs:224-527
//   weights(code, values)              /root/reference/src/groth16/circuit/mod.rs:529-637
//   evaluate                           /root/reference/src/groth16/circuit/mod.rs:639-656
//   legacy whitespace format (Z251)    /root/reference/src/groth16/circuit/dummy_rep.rs:55-141
// Pinned by the reference's parser KATs (circuit/mod.rs:664-769), see tests/test_oracle_kats.py.
//
// Rust `Result::Err(ParseErr)` is modelled by throwing ParseErr; Rust panics by Panic.
#pragma once
#include <map>
#include <sstream>
#include <string>
#include <unordered_map>
#include <vector>
#include "groth16.hpp"

namespace orc {

struct ParseErr {
    enum Kind { SyntaxErr, StructureErr } kind;
    long where;  // line (SyntaxErr) or gate number (StructureErr); -1 == None
    std::string msg;
};
struct Panic : std::runtime_error { using std::runtime_error::runtime_error; };

enum class Key { In, Out, Verify, Program, Equal, Mul, Add };
template <class T>
struct Token {
    enum Kind { Keyword, Var, Open, Close, Literal } kind;
    Key key{};
    std::string var;
    T lit{};
};

template <class T>
struct Expr {
    enum Kind { In, Out, Verify, Program, Assign, Mul, Add, Var, Literal } kind;
    std::vector<Expr> kids;
    std::string var;
    T lit{};
};

static inline std::vector<std::string> split_lines(const std::string& code) {
    // str::lines(): split on '\n', strip one trailing '\r', no trailing empty line
    std::vector<std::string> out;
    size_t i = 0;
    while (i < code.size()) {
        size_t j = code.find('\n', i);
        if (j == std::string::npos) j = code.size();
        std::string ln = code.substr(i, j - i);
        if (!ln.empty() && ln.back() == '\r') ln.pop_back();
        out.push_back(ln);
        i = j + 1;
    }
    return out;
}

// ast.rs:300-370
template <class T>
void parse_token(std::string substr, std::vector<Token<T>>& tokens, long line) {
    size_t first_len = tokens.size();
    bool opened = false;
    if (!substr.empty() && substr[0] == '(') {
        Token<T> t; t.kind = Token<T>::Open; tokens.push_back(t);
        substr = substr.substr(1);
        opened = true;
    }
    if (substr.empty()) throw ParseErr{ParseErr::SyntaxErr, line, "found whitespace after '('"};
    static const std::map<std::string, Key> kw = {{"in", Key::In}, {"out", Key::Out}, {"verify", Key::Verify},
                                                  {"program", Key::Program}, {"=", Key::Equal}, {"*", Key::Mul}, {"+", Key::Add}};
    auto it = kw.find(substr);
    if (it != kw.end()) {
        Token<T> t; t.kind = Token<T>::Keyword; t.key = it->second; tokens.push_back(t);
        return;
    }
    if (substr.find('(') != std::string::npos) throw ParseErr{ParseErr::SyntaxErr, line, "unexpected '('"};
    if (substr.find_first_of("*+=") != std::string::npos) throw ParseErr{ParseErr::SyntaxErr, line, "unexpected operator"};
    size_t cut = substr.find(')');
    std::string start = cut == std::string::npos ? substr : substr.substr(0, cut);
    std::string end = cut == std::string::npos ? "" : substr.substr(cut);
    if (opened && !end.empty()) throw ParseErr{ParseErr::SyntaxErr, line, "unexpected ')'"};
    if (start.empty()) throw Panic("called `Option::unwrap()` on a `None` value");  // ast.rs:347
    unsigned char f = (unsigned char)start[0];
    if (f >= '0' && f <= '9') {  // char::is_numeric on ASCII input
        T lit;
        if (!T::from_str(start, lit)) throw ParseErr{ParseErr::SyntaxErr, line, "could not parse literal"};
        Token<T> t; t.kind = Token<T>::Literal; t.lit = lit; tokens.push_back(t);
    } else {
        Token<T> t; t.kind = Token<T>::Var; t.var = start; tokens.push_back(t);
    }
    for (char c : end) {
        if (c != ')') throw ParseErr{ParseErr::SyntaxErr, line, "expected ')'"};
        Token<T> t; t.kind = Token<T>::Close; tokens.push_back(t);
    }
    (void)first_len;
}

// ast.rs:263-287
template <class T>
std::vector<Token<T>> try_to_list(const std::string& code) {
    std::vector<Token<T>> tokens;
    long line = 1;
    for (const auto& ln : split_lines(code)) {
        std::istringstream ss(ln);
        std::string w;
        while (ss >> w) parse_token<T>(w, tokens, line);
        ++line;
    }
    return tokens;
}

// ast.rs:230-261.  `pos` is the shared iterator.
template <class T>
std::vector<Token<T>> next_group(const std::vector<Token<T>>& toks, size_t& pos) {
    std::vector<Token<T>> out;
    if (pos >= toks.size()) return out;
    const Token<T>& t = toks[pos++];
    if (t.kind == Token<T>::Open) {
        int depth = 1;
        while (pos < toks.size()) {
            const Token<T>& u = toks[pos++];
            if (u.kind == Token<T>::Open) ++depth;
            else if (u.kind == Token<T>::Close) --depth;
            if (depth == 0) break;
            out.push_back(u);
        }
        return out;
    }
    if (t.kind == Token<T>::Var || t.kind == Token<T>::Literal) { out.push_back(t); return out; }
    throw Panic("Cannot parse malformed group");
}

// ast.rs:106-228
template <class T>
Expr<T> parse_expression(const std::vector<Token<T>>& toks) {
    typedef Expr<T> E;
    size_t pos = 0;
    if (toks.empty()) throw ParseErr{ParseErr::StructureErr, -1, "Malformed expression"};
    const Token<T>& head = toks[pos++];
    E e;
    if (head.kind == Token<T>::Var) { e.kind = E::Var; e.var = head.var; return e; }
    if (head.kind == Token<T>::Literal) { e.kind = E::Literal; e.lit = head.lit; return e; }
    if (head.kind != Token<T>::Keyword) throw ParseErr{ParseErr::StructureErr, -1, "Malformed expression"};
    switch (head.key) {
        case Key::In: case Key::Out: case Key::Verify: {
            e.kind = head.key == Key::In ? E::In : head.key == Key::Out ? E::Out : E::Verify;
            const char* nm = head.key == Key::In ? "in" : head.key == Key::Out ? "out" : "verify";
            for (; pos < toks.size(); ++pos) {
                if (toks[pos].kind != Token<T>::Var)
                    throw ParseErr{ParseErr::StructureErr, -1, std::string("Non variable found in '") + nm + "' expression"};
                E v; v.kind = E::Var; v.var = toks[pos].var; e.kids.push_back(v);
            }
            return e;
        }
        case Key::Program: case Key::Add: {
            e.kind = head.key == Key::Program ? E::Program : E::Add;
            for (;;) {
                auto g = next_group(toks, pos);
                if (g.empty()) break;
                e.kids.push_back(parse_expression(g));
            }
            return e;
        }
        case Key::Equal: {
            auto left = next_group(toks, pos);
            if (left.size() != 1 || left[0].kind != Token<T>::Var)
                throw ParseErr{ParseErr::StructureErr, -1, "Can only assign to a variable"};
            E l; l.kind = E::Var; l.var = left[0].var;
            E r = parse_expression(next_group(toks, pos));
            e.kind = E::Assign; e.kids = {l, r};
            return e;
        }
        case Key::Mul: {
            E l = parse_expression(next_group(toks, pos));
            E r = parse_expression(next_group(toks, pos));
            e.kind = E::Mul; e.kids = {l, r};
            return e;
        }
    }
    throw ParseErr{ParseErr::StructureErr, -1, "Malformed expression"};
}

// ast.rs:85-104
template <class T>
std::vector<Expr<T>> expressions(const std::string& code) {
    auto toks = try_to_list<T>(code);
    size_t pos = 0;
    std::vector<Expr<T>> out;
    for (;;) {
        auto g = next_group(toks, pos);
        if (g.empty()) break;
        out.push_back(parse_expression(g));
    }
    return out;
}

// ast.rs:62-83
template <class T>
std::vector<std::string> variable_order(const std::vector<Token<T>>& toks) {
    std::vector<std::string> out;
    std::unordered_map<std::string, bool> seen;
    size_t i = 0;
    while (i < toks.size() && !(toks[i].kind == Token<T>::Keyword && toks[i].key == Key::Verify)) ++i;
    for (; i < toks.size(); ++i)
        if (toks[i].kind == Token<T>::Var && !seen.count(toks[i].var)) { seen[toks[i].var] = true; out.push_back(toks[i].var); }
    return out;
}

// circuit/mod.rs:224-527
template <class F>
DummyRep<F> ast_try_parse(const std::string& code) {
    typedef Expr<F> E;
    auto exps = expressions<F>(code);
    std::unordered_map<std::string, size_t> variables;
    size_t gate_number = 0;
    typedef std::vector<std::pair<F, F>> Row;
    std::vector<Row> u(1), v(1), w(1);
    size_t input = 0;
    auto serr = [&](const std::string& m) { return ParseErr{ParseErr::StructureErr, (long)gate_number, m}; };
    if (exps.size() != 4) throw serr("Expected exactly one each of 'in', 'out', 'verify' and 'program'");
    if (exps[0].kind != E::In) throw serr("Expected first expression to be 'in'");
    if (exps[1].kind != E::Out) throw serr("Expected second expression to be 'out'");
    if (exps[2].kind != E::Verify) throw serr("Expected third expression to be 'verify'");
    for (const auto& var : exps[2].kids) {
        variables[var.var] = u.size();  // HashMap::insert overwrites on duplicates
        u.emplace_back(); v.emplace_back(); w.emplace_back();
        ++input;
    }
    if (exps[3].kind != E::Program) throw serr("Expected fourth expression to be 'program'");
    auto gate = [&]() { return F::from_usize(gate_number); };
    auto one = [&]() { return F::from_usize(1); };
    // adds (gate, coeff) for variable `name` on side `side` (u or v), creating the wire if new
    auto touch = [&](std::vector<Row>& side, const std::string& name, const F& coeff) {
        auto it = variables.find(name);
        if (it == variables.end()) {
            variables[name] = side.size();
            u.emplace_back(); v.emplace_back(); w.emplace_back();
            side.back().push_back({gate(), coeff});
        } else {
            side[it->second].push_back({gate(), coeff});
        }
    };
    auto side_input = [&](std::vector<Row>& side, const E& ex) {
        switch (ex.kind) {
            case E::Literal: side[0].push_back({gate(), ex.lit}); break;
            case E::Var: touch(side, ex.var, one()); break;
            case E::Add:
                for (const auto& t : ex.kids) {
                    if (t.kind == E::Literal) side[0].push_back({gate(), t.lit});
                    else if (t.kind == E::Var) touch(side, t.var, one());
                    else if (t.kind == E::Mul) {
                        if (t.kids[0].kind != E::Literal) throw serr("LHS of a '*' expression in a '+' expression must be a literal");
                        if (t.kids[1].kind != E::Var) throw serr("RHS of a '*' expression in a '+' expression must be a variable");
                        touch(side, t.kids[1].var, t.kids[0].lit);
                    } else throw serr("Invalid expression found in '+' expression");
                }
                break;
            default: throw serr("Invalid expression found in '*' expression");
        }
    };
    for (const auto& asg : exps[3].kids) {
        ++gate_number;
        if (asg.kind != E::Assign) throw serr("Program expression must be a list of '=' expressions");
        const std::string& out = asg.kids[0].var;
        auto it = variables.find(out);
        if (it == variables.end()) {
            variables[out] = u.size();
            u.emplace_back(); v.emplace_back(); w.emplace_back();
            w.back().push_back({gate(), one()});
        } else if (it->second <= input) {
            if (!w[it->second].empty()) throw serr("Varify variable cannot be the output of two different gates");
            w[it->second].push_back({gate(), one()});
        } else {
            throw serr("Already declared variable cannot be the output wire of a gate");
        }
        const E& rhs = asg.kids[1];
        if (rhs.kind == E::Mul) {  // a non-Mul right-hand side is silently accepted (mod.rs:337)
            side_input(u, rhs.kids[0]);
            side_input(v, rhs.kids[1]);
        }
    }
    DummyRep<F> rep;
    rep.u = u; rep.v = v; rep.w = w; rep.input = input;
    for (size_t k = 1; k <= gate_number; ++k) rep.roots.push_back(F::from_usize(k));
    return rep;
}

// circuit/mod.rs:639-656; returns false for None
template <class F>
bool zk_evaluate(const Expr<F>& e, const std::unordered_map<std::string, F>& asg, F& out) {
    typedef Expr<F> E;
    switch (e.kind) {
        case E::Literal: out = e.lit; return true;
        case E::Var: { auto it = asg.find(e.var); if (it == asg.end()) return false; out = it->second; return true; }
        case E::Mul: {
            F l, r;
            if (!zk_evaluate(e.kids[0], asg, l)) return false;
            if (!zk_evaluate(e.kids[1], asg, r)) return false;
            out = l * r; return true;
        }
        case E::Add: {
            F acc = F::zero();
            for (const auto& k : e.kids) { F t; if (!zk_evaluate(k, asg, t)) return false; acc = acc + t; }
            out = acc; return true;
        }
        default: return false;
    }
}

// circuit/mod.rs:529-637
template <class F>
std::vector<F> zk_weights(const std::string& code, const std::vector<F>& values) {
    typedef Expr<F> E;
    std::unordered_map<std::string, F> asg;
    auto exps = expressions<F>(code);
    auto order = variable_order(try_to_list<F>(code));
    auto serr = [&](const std::string& m) { return ParseErr{ParseErr::StructureErr, -1, m}; };
    if (exps.size() < 1 || exps[0].kind != E::In) throw serr("Expected first expression to be 'in'");
    if (exps[0].kids.size() != values.size()) throw serr("Wrong number of values supplied");
    for (size_t i = 0; i < values.size(); ++i) asg[exps[0].kids[i].var] = values[i];
    if (exps.size() < 2 || exps[1].kind != E::Out) throw serr("Expected second expression to be 'out'");
    if (exps.size() < 3 || exps[2].kind != E::Verify) throw serr("Expected third expression to be 'verify'");
    if (exps.size() < 4 || exps[3].kind != E::Program) throw serr("Expected fourth expression to be 'program'");
    for (const auto& a : exps[3].kids) {
        if (a.kind != E::Assign) throw serr("Program expression must be a list of '=' expressions");
        const std::string& var = a.kids[0].var;
        if (asg.count(var)) throw serr("Attempted to assign to an already assigned variable");
        F val;
        if (!zk_evaluate(a.kids[1], asg, val)) throw serr("Under constrained expression");
        asg[var] = val;
    }
    std::vector<F> out{F::one()};
    for (const auto& name : order) {
        auto it = asg.find(name);
        if (it == asg.end()) throw Panic("Every variable should have an assignment");
        out.push_back(it->second);
        asg.erase(it);
    }
    return out;
}

// dummy_rep.rs:55-141 (legacy whitespace format, Z251 only in the reference)
template <class F>
DummyRep<F> legacy_parse(const std::string& code) {
    auto lines = split_lines(code);
    auto split_sp = [](const std::string& s) {
        std::vector<std::string> out; size_t i = 0;
        for (;;) { size_t j = s.find(' ', i); if (j == std::string::npos) { out.push_back(s.substr(i)); break; } out.push_back(s.substr(i, j - i)); i = j + 1; }
        return out;
    };
    if (lines.size() < 3) throw Panic("called `Option::unwrap()` on a `None` value");
    auto inputs = split_sp(lines[0]), witness = split_sp(lines[1]), temps = split_sp(lines[2]);
    std::vector<std::string> all = inputs;
    all.insert(all.end(), witness.begin(), witness.end());
    all.insert(all.end(), temps.begin(), temps.end());
    size_t num_vars = all.size() + 1;
    DummyRep<F> rep;
    rep.u.resize(num_vars); rep.v.resize(num_vars); rep.w.resize(num_vars);
    auto position = [&](const std::string& s) {
        for (size_t i = 0; i < all.size(); ++i) if (all[i] == s) return i + 1;
        throw Panic("called `Option::unwrap()` on a `None` value");
    };
    size_t line_count = 0;
    for (size_t li = 4; li < lines.size(); ++li) {
        size_t n = line_count++;
        auto sym = split_sp(lines[li]);
        size_t p = 0;
        rep.w[position(sym.at(p++))].push_back({F::from_usize(n + 1), F::from_usize(1)});
        ++p;  // "("
        for (; p < sym.size() && sym[p] != ")"; ++p) {
            if (sym[p] == "1") rep.u[0].push_back({F::from_usize(n + 1), F::from_usize(1)});
            else rep.u[position(sym[p])].push_back({F::from_usize(n + 1), F::from_usize(1)});
        }
        ++p;  // ")" consumed by take_while
        ++p;  // "("
        for (; p < sym.size() && sym[p] != ")"; ++p)
            rep.v[position(sym[p])].push_back({F::from_usize(n + 1), F::from_usize(1)});
    }
    for (size_t k = 1; k <= line_count; ++k) rep.roots.push_back(F::from_usize(k));
    rep.input = inputs.size();
    return rep;
}

}  // namespace orc
