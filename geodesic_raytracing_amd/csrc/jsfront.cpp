// jsfront.cpp — metric-script front-end: a small tree-walking interpreter for the JavaScript dialect the reference's
// metric scripts are written in, evaluating them over symbolic numbers (sym::E / sym::Cx).
//
// Reference counterpart: js_interop.{hpp,cpp} + number.js (QuickJS with operator overloading routed to dual numbers)
// and content_manager.cpp:9-112 (JSON config with one level of inherit_settings, lookup of the to_polar / from_polar /
// origin_distance / coordinate_periodicity scripts by file name).  QuickJS and deps/vec are not part of the reference
// checkout; this is an independent implementation of the same script API:
//   * the script's completion value is the metric function `(v1,v2,v3,v4) -> [4] | [16]`         js_interop.cpp:848-901
//   * + - * / and unary - on {number, real symbolic, complex symbolic}                             number.js:12-73
//   * CMath.{sin,cos,tan,asin,acos,atan,atan2,fabs,log,exp,sqrt,psqrt,csqrt,pow,sinh,cosh,tanh,select,lt,lte,eq,gt,gte,
//     fast_length,length,smooth_fmod,conjugate,self_conjugate_multiply,Real,Imaginary,i,M_PI,PI}  js_interop.cpp:690-732
//   * $cfg.NAME (run-time parameter, registered on first mention) and $cfg.NAME.$default = x       js_interop.cpp:762-815
//   * $pin(x): accepted; common sub-expressions are hoisted automatically (sym::hoist_position_temporaries)
//   * Math.PI, M_PI
// Language subset: function declarations/expressions with closures, var (function scoped) / let / const, numbers,
// strings, array literals, index and .length (get/set), if/else, for, while, return, ?:, comparison and logical
// operators on plain numbers, ++/--, compound assignment, comments, automatic semicolon insertion at line ends.
#include "jsfront.hpp"

#include <cmath>
#include <cstring>
#include <fstream>
#include <functional>
#include <sstream>
#include <stdexcept>
#include <sys/stat.h>

#include "builtin_metrics.hpp"

using namespace sym;

namespace gr {
namespace {

[[noreturn]] void fail(const std::string& msg) { throw std::runtime_error(msg); }

// ------------------------------------------------------------------------------------------------
// lexer

enum TokKind { T_NUM, T_STR, T_IDENT, T_PUNCT, T_EOF };
struct Token {
    TokKind kind;
    std::string text;
    double num = 0;
    bool newline_before = false;
    int line = 0;
};

std::vector<Token> lex(const std::string& src) {
    std::vector<Token> out;
    size_t i = 0, n = src.size();
    int line = 1;
    bool nl = false;
    auto push = [&](TokKind k, const std::string& t, double v = 0) {
        Token tk;
        tk.kind = k; tk.text = t; tk.num = v; tk.newline_before = nl; tk.line = line;
        out.push_back(tk);
        nl = false;
    };
    while (i < n) {
        char c = src[i];
        if (c == '\n') { nl = true; line++; i++; continue; }
        if (c == ' ' || c == '\t' || c == '\r') { i++; continue; }
        if (c == '/' && i + 1 < n && src[i + 1] == '/') { while (i < n && src[i] != '\n') i++; continue; }
        if (c == '/' && i + 1 < n && src[i + 1] == '*') {
            i += 2;
            while (i + 1 < n && !(src[i] == '*' && src[i + 1] == '/')) { if (src[i] == '\n') { line++; nl = true; } i++; }
            i += 2;
            continue;
        }
        if (isdigit((unsigned char)c) || (c == '.' && i + 1 < n && isdigit((unsigned char)src[i + 1]))) {
            size_t j = i;
            while (j < n && isdigit((unsigned char)src[j])) j++;
            if (j < n && src[j] == '.') { j++; while (j < n && isdigit((unsigned char)src[j])) j++; }
            if (j < n && (src[j] == 'e' || src[j] == 'E')) {
                size_t k = j + 1;
                if (k < n && (src[k] == '+' || src[k] == '-')) k++;
                if (k < n && isdigit((unsigned char)src[k])) { j = k; while (j < n && isdigit((unsigned char)src[j])) j++; }
            }
            push(T_NUM, src.substr(i, j - i), std::strtod(src.substr(i, j - i).c_str(), nullptr));
            i = j;
            continue;
        }
        if (isalpha((unsigned char)c) || c == '_' || c == '$') {
            size_t j = i;
            while (j < n && (isalnum((unsigned char)src[j]) || src[j] == '_' || src[j] == '$')) j++;
            push(T_IDENT, src.substr(i, j - i));
            i = j;
            continue;
        }
        if (c == '"' || c == '\'') {
            size_t j = i + 1;
            std::string s;
            while (j < n && src[j] != c) {
                if (src[j] == '\\' && j + 1 < n) { j++; s += src[j] == 'n' ? '\n' : src[j] == 't' ? '\t' : src[j]; }
                else s += src[j];
                j++;
            }
            push(T_STR, s);
            i = j + 1;
            continue;
        }
        static const char* ops[] = {"===", "!==", "==", "!=", "<=", ">=", "&&", "||", "++", "--", "+=", "-=", "*=", "/=", "=>"};
        bool matched = false;
        for (const char* op : ops) {
            size_t l = strlen(op);
            if (src.compare(i, l, op) == 0) { push(T_PUNCT, op); i += l; matched = true; break; }
        }
        if (matched) continue;
        push(T_PUNCT, std::string(1, c));
        i++;
    }
    push(T_EOF, "");
    return out;
}

// ------------------------------------------------------------------------------------------------
// AST

struct Node;
typedef std::shared_ptr<Node> NodeP;
enum NodeKind {
    N_NUM, N_STR, N_IDENT, N_ARRAY, N_OBJECT, N_MEMBER, N_INDEX, N_CALL, N_UNARY, N_BINARY, N_ASSIGN, N_COND, N_FUNC, N_UPDATE,
    S_VAR, S_EXPR, S_RETURN, S_IF, S_FOR, S_WHILE, S_BLOCK, S_FUNCDECL, S_EMPTY
};
struct Node {
    NodeKind kind;
    std::string text;           // identifier / operator / property
    double num = 0;
    std::vector<NodeP> kids;    // operands / statements
    std::vector<std::string> params;
    std::vector<std::pair<std::string, NodeP>> decls;   // var declarations
    bool prefix = false, block_scoped = false;
    int line = 0;
};

struct Parser {
    std::vector<Token> t;
    size_t p = 0;
    explicit Parser(const std::string& src) : t(lex(src)) {}

    const Token& peek() const { return t[p]; }
    bool is(const char* s) const { return t[p].kind == T_PUNCT && t[p].text == s; }
    bool is_ident(const char* s) const { return t[p].kind == T_IDENT && t[p].text == s; }
    bool accept(const char* s) { if (is(s)) { p++; return true; } return false; }
    void expect(const char* s) { if (!accept(s)) fail("script line " + std::to_string(peek().line) + ": expected '" + s + "' near '" + peek().text + "'"); }
    NodeP mk(NodeKind k) { auto n = std::make_shared<Node>(); n->kind = k; n->line = peek().line; return n; }

    void end_statement() {
        if (accept(";")) return;
        if (peek().kind == T_EOF || is("}") || peek().newline_before) return;   // automatic semicolon insertion
        fail("script line " + std::to_string(peek().line) + ": unexpected '" + peek().text + "'");
    }

    NodeP program() {
        auto b = mk(S_BLOCK);
        while (peek().kind != T_EOF) b->kids.push_back(statement());
        return b;
    }

    NodeP function_rest(const std::string& name) {
        auto f = mk(N_FUNC);
        f->text = name;
        expect("(");
        while (!is(")")) {
            if (peek().kind != T_IDENT) fail("bad parameter list");
            f->params.push_back(t[p++].text);
            if (!accept(",")) break;
        }
        expect(")");
        f->kids.push_back(block());
        return f;
    }

    NodeP block() {
        expect("{");
        auto b = mk(S_BLOCK);
        while (!is("}")) {
            if (peek().kind == T_EOF) fail("unterminated block");
            b->kids.push_back(statement());
        }
        expect("}");
        return b;
    }

    NodeP statement() {
        if (accept(";")) return mk(S_EMPTY);
        if (is("{")) return block();
        if (is_ident("function") && t[p + 1].kind == T_IDENT) {
            p++;
            std::string name = t[p++].text;
            auto d = mk(S_FUNCDECL);
            d->text = name;
            d->kids.push_back(function_rest(name));
            return d;
        }
        if (is_ident("var") || is_ident("let") || is_ident("const")) {
            auto v = mk(S_VAR);
            v->block_scoped = !is_ident("var");
            p++;
            do {
                if (peek().kind != T_IDENT) fail("bad declaration");
                std::string name = t[p++].text;
                NodeP init;
                if (accept("=")) init = assignment();
                v->decls.emplace_back(name, init);
            } while (accept(","));
            end_statement();
            return v;
        }
        if (is_ident("return")) {
            auto r = mk(S_RETURN);
            p++;
            if (!is(";") && !is("}") && peek().kind != T_EOF && !peek().newline_before) r->kids.push_back(expression());
            end_statement();
            return r;
        }
        if (is_ident("if")) {
            auto s = mk(S_IF);
            p++;
            expect("(");
            s->kids.push_back(expression());
            expect(")");
            s->kids.push_back(statement());
            if (is_ident("else")) { p++; s->kids.push_back(statement()); }
            return s;
        }
        if (is_ident("for")) {
            auto s = mk(S_FOR);
            p++;
            expect("(");
            s->kids.push_back(is(";") ? (p++, mk(S_EMPTY)) : statement());     // init (consumes its ';')
            s->kids.push_back(is(";") ? nullptr : expression());
            expect(";");
            s->kids.push_back(is(")") ? nullptr : expression());
            expect(")");
            s->kids.push_back(statement());
            return s;
        }
        if (is_ident("while")) {
            auto s = mk(S_WHILE);
            p++;
            expect("(");
            s->kids.push_back(expression());
            expect(")");
            s->kids.push_back(statement());
            return s;
        }
        auto e = mk(S_EXPR);
        e->kids.push_back(expression());
        end_statement();
        return e;
    }

    NodeP expression() {
        NodeP e = assignment();
        while (accept(",")) e = assignment();   // comma operator: value of the last
        return e;
    }

    NodeP assignment() {
        NodeP lhs = conditional();
        for (const char* op : {"=", "+=", "-=", "*=", "/="}) {
            if (is(op)) {
                auto a = mk(N_ASSIGN);
                a->text = op;
                p++;
                a->kids = {lhs, assignment()};
                return a;
            }
        }
        return lhs;
    }

    NodeP conditional() {
        NodeP c = binary(0);
        if (accept("?")) {
            auto n = mk(N_COND);
            NodeP a = assignment();
            expect(":");
            NodeP b = assignment();
            n->kids = {c, a, b};
            return n;
        }
        return c;
    }

    static int precedence(const std::string& op) {
        if (op == "||") return 1;
        if (op == "&&") return 2;
        if (op == "==" || op == "!=" || op == "===" || op == "!==") return 3;
        if (op == "<" || op == ">" || op == "<=" || op == ">=") return 4;
        if (op == "+" || op == "-") return 5;
        if (op == "*" || op == "/" || op == "%") return 6;
        return -1;
    }

    NodeP binary(int min_prec) {
        NodeP lhs = unary();
        while (peek().kind == T_PUNCT) {
            int prec = precedence(peek().text);
            if (prec < 0 || prec < min_prec) break;
            auto b = mk(N_BINARY);
            b->text = t[p++].text;
            NodeP rhs = binary(prec + 1);
            b->kids = {lhs, rhs};
            lhs = b;
        }
        return lhs;
    }

    NodeP unary() {
        if (is("-") || is("+") || is("!")) {
            auto u = mk(N_UNARY);
            u->text = t[p++].text;
            u->kids.push_back(unary());
            return u;
        }
        if (is("++") || is("--")) {
            auto u = mk(N_UPDATE);
            u->text = t[p++].text;
            u->prefix = true;
            u->kids.push_back(unary());
            return u;
        }
        return postfix();
    }

    NodeP postfix() {
        NodeP e = primary();
        for (;;) {
            if (accept(".")) {
                auto m = mk(N_MEMBER);
                if (peek().kind != T_IDENT) fail("bad member access");
                m->text = t[p++].text;
                m->kids.push_back(e);
                e = m;
            } else if (is("[")) {
                p++;
                auto m = mk(N_INDEX);
                m->kids = {e, expression()};
                expect("]");
                e = m;
            } else if (is("(") ) {
                p++;
                auto c = mk(N_CALL);
                c->kids.push_back(e);
                while (!is(")")) {
                    c->kids.push_back(assignment());
                    if (!accept(",")) break;
                }
                expect(")");
                e = c;
            } else if ((is("++") || is("--")) && !peek().newline_before) {
                auto u = mk(N_UPDATE);
                u->text = t[p++].text;
                u->kids.push_back(e);
                e = u;
            } else {
                break;
            }
        }
        return e;
    }

    NodeP primary() {
        const Token& tk = peek();
        if (tk.kind == T_NUM) { auto n = mk(N_NUM); n->num = tk.num; p++; return n; }
        if (tk.kind == T_STR) { auto n = mk(N_STR); n->text = tk.text; p++; return n; }
        if (tk.kind == T_IDENT) {
            if (tk.text == "function") {
                p++;
                std::string name;
                if (peek().kind == T_IDENT) name = t[p++].text;
                return function_rest(name);
            }
            auto n = mk(N_IDENT);
            n->text = tk.text;
            p++;
            return n;
        }
        if (accept("(")) { NodeP e = expression(); expect(")"); return e; }
        if (accept("[")) {
            auto a = mk(N_ARRAY);
            while (!is("]")) {
                a->kids.push_back(assignment());
                if (!accept(",")) break;
            }
            expect("]");
            return a;
        }
        if (is("{")) {
            p++;
            auto o = mk(N_OBJECT);
            while (!is("}")) {
                if (peek().kind != T_IDENT && peek().kind != T_STR) fail("bad object literal");
                std::string key = t[p++].text;
                expect(":");
                o->decls.emplace_back(key, assignment());
                if (!accept(",")) break;
            }
            expect("}");
            return o;
        }
        fail("script line " + std::to_string(tk.line) + ": unexpected '" + tk.text + "'");
    }
};

// ------------------------------------------------------------------------------------------------
// values

struct Value;
struct Env;
typedef std::shared_ptr<Env> EnvP;
struct Closure {
    NodeP fn;
    EnvP env;
};
typedef std::function<Value(std::vector<Value>&)> Native;

enum VKind { V_UNDEF, V_BOOL, V_NUM, V_STR, V_SYM, V_CPX, V_ARR, V_OBJ, V_FUNC, V_NATIVE, V_CFG };
struct Value {
    VKind k = V_UNDEF;
    bool b = false;
    double num = 0;
    std::string str;      // string value; for V_SYM created by $cfg: the parameter name
    E e = nullptr;
    Cx c{nullptr, nullptr};
    std::shared_ptr<std::vector<Value>> arr;
    std::shared_ptr<std::map<std::string, Value>> obj;
    std::shared_ptr<Closure> fn;
    std::shared_ptr<Native> native;

    static Value number(double v) { Value x; x.k = V_NUM; x.num = v; return x; }
    static Value boolean(bool v) { Value x; x.k = V_BOOL; x.b = v; return x; }
    static Value symbol(E e) { Value x; x.k = V_SYM; x.e = e; return x; }
    static Value complex(Cx c) { Value x; x.k = V_CPX; x.c = c; return x; }
    static Value array() { Value x; x.k = V_ARR; x.arr = std::make_shared<std::vector<Value>>(); return x; }
    static Value object() { Value x; x.k = V_OBJ; x.obj = std::make_shared<std::map<std::string, Value>>(); return x; }
    static Value nat(Native f) { Value x; x.k = V_NATIVE; x.native = std::make_shared<Native>(std::move(f)); return x; }
};

struct Env {
    std::map<std::string, Value> vars;
    EnvP parent;
    bool is_function = false;
    Value* find(const std::string& n) {
        for (Env* e = this; e; e = e->parent.get()) {
            auto it = e->vars.find(n);
            if (it != e->vars.end()) return &it->second;
        }
        return nullptr;
    }
    Env* function_scope() {
        Env* e = this;
        while (e->parent && !e->is_function) e = e->parent.get();
        return e;
    }
};

bool is_numeric(const Value& v) { return v.k == V_NUM || v.k == V_BOOL || v.k == V_SYM || v.k == V_CPX || v.k == V_UNDEF; }
bool is_plain(const Value& v) { return v.k == V_NUM || v.k == V_BOOL; }
double plain(const Value& v) { return v.k == V_BOOL ? (v.b ? 1.0 : 0.0) : v.num; }

// real symbolic view of a value (storage::d in the reference; undefined reads as 0 like an empty storage)
E as_real(const Value& v) {
    switch (v.k) {
        case V_NUM: return constant(v.num);
        case V_BOOL: return constant(v.b ? 1.0 : 0.0);
        case V_SYM: return v.e;
        case V_CPX: return v.c.re;                  // getr(): Real(s.c), js_interop.cpp:382-392
        case V_UNDEF: return constant(0.0);
        default: fail("expected a number");
    }
}
Cx as_complex(const Value& v) { return v.k == V_CPX ? v.c : cx(as_real(v)); }

bool truthy(const Value& v) {
    switch (v.k) {
        case V_UNDEF: return false;
        case V_BOOL: return v.b;
        case V_NUM: return v.num != 0 && !std::isnan(v.num);
        case V_STR: return !v.str.empty();
        case V_SYM:
            if (is_const(v.e)) return v.e->c != 0;
            fail("a symbolic value cannot be used as a condition (use CMath.select)");
        default: return true;
    }
}

Value arith(const std::string& op, const Value& a, const Value& b) {
    if (op == "+" && (a.k == V_STR || b.k == V_STR)) {
        auto str = [](const Value& v) { return v.k == V_STR ? v.str : v.k == V_NUM ? std::to_string(v.num) : std::string("?"); };
        Value s; s.k = V_STR; s.str = str(a) + str(b);
        return s;
    }
    if (!is_numeric(a) || !is_numeric(b)) fail("arithmetic on a non-number");
    if (is_plain(a) && is_plain(b)) {
        double x = plain(a), y = plain(b);
        if (op == "+") return Value::number(x + y);
        if (op == "-") return Value::number(x - y);
        if (op == "*") return Value::number(x * y);
        if (op == "/") return Value::number(x / y);
        if (op == "%") return Value::number(std::fmod(x, y));
    }
    if (a.k == V_CPX || b.k == V_CPX) {
        Cx x = as_complex(a), y = as_complex(b);
        if (op == "+") return Value::complex(cadd(x, y));
        if (op == "-") return Value::complex(csub(x, y));
        if (op == "*") return Value::complex(cmul(x, y));
        if (op == "/") return Value::complex(cdiv(x, y));
        fail("operator " + op + " is not defined on complex values");
    }
    E x = as_real(a), y = as_real(b);
    if (op == "+") return Value::symbol(add(x, y));
    if (op == "-") return Value::symbol(sub(x, y));
    if (op == "*") return Value::symbol(mul(x, y));
    if (op == "/") return Value::symbol(div(x, y));
    if (op == "%") return Value::symbol(fn2(F_FMOD, x, y));
    fail("bad operator " + op);
}

struct ReturnSignal { Value v; };

// ------------------------------------------------------------------------------------------------
// interpreter

struct Interpreter {
    DynamicVars* vars;
    EnvP global;
    Value completion;

    explicit Interpreter(DynamicVars* dv) : vars(dv) {
        global = std::make_shared<Env>();
        global->is_function = true;
        install_host();
    }

    static Value real_fn(Fn f, const Value& a) { return is_plain(a) ? Value::symbol(fn1(f, constant(plain(a)))) : Value::symbol(fn1(f, as_real(a))); }

    void install_host() {
        Value cmath = Value::object();
        auto& m = *cmath.obj;
        auto unary_real = [](Fn f) {
            return Value::nat([f](std::vector<Value>& a) {
                if (a.size() < 1) fail("CMath function needs an argument");
                if (a[0].k == V_CPX) fail("this CMath function is real-only");
                return Value::symbol(fn1(f, as_real(a[0])));
            });
        };
        for (auto& [name, f] : std::vector<std::pair<std::string, Fn>>{{"tan", F_TAN}, {"asin", F_ASIN}, {"acos", F_ACOS}, {"atan", F_ATAN},
                                                                       {"log", F_LOG}, {"exp", F_EXP}, {"sinh", F_SINH}, {"cosh", F_COSH}, {"tanh", F_TANH}})
            m[name] = unary_real(f);
        m["sin"] = Value::nat([](std::vector<Value>& a) { return a.at(0).k == V_CPX ? Value::complex(csin(a[0].c)) : Value::symbol(fn1(F_SIN, as_real(a[0]))); });
        m["cos"] = Value::nat([](std::vector<Value>& a) { return a.at(0).k == V_CPX ? Value::complex(ccos(a[0].c)) : Value::symbol(fn1(F_COS, as_real(a[0]))); });
        m["fabs"] = Value::nat([](std::vector<Value>& a) { return a.at(0).k == V_CPX ? Value::symbol(cabs(a[0].c)) : Value::symbol(fn1(F_FABS, as_real(a[0]))); });
        // sqrt / psqrt: real argument -> real sqrt, complex argument -> principal complex root (deps/vec dual_complex;
        // branch convention inferred from the names)
        auto sqrt_fn = Value::nat([](std::vector<Value>& a) { return a.at(0).k == V_CPX ? Value::complex(csqrt_principal(a[0].c)) : Value::symbol(fn1(F_SQRT, as_real(a[0]))); });
        m["sqrt"] = sqrt_fn;
        m["psqrt"] = sqrt_fn;
        m["csqrt"] = Value::nat([](std::vector<Value>& a) {
            if (a.at(0).k == V_CPX) fail("csqrt must be used with purely real arguments");
            return Value::complex(csqrt_real(as_real(a[0])));
        });
        m["atan2"] = Value::nat([](std::vector<Value>& a) { return Value::symbol(fn2(F_ATAN2, as_real(a.at(0)), as_real(a.at(1)))); });
        m["pow"] = Value::nat([](std::vector<Value>& a) {
            if (a.size() < 2 || a[1].k == V_CPX) fail("Pow cannot be used with a complex second argument");
            E y = as_real(a[1]);
            if (a[0].k == V_CPX) {
                if (!is_const(y) || y->c != std::floor(y->c)) fail("With a complex first argument, the exponent must be a constant integer");
                return Value::complex(cpowi(a[0].c, (int)y->c));
            }
            return Value::symbol(fn2(F_POW, as_real(a[0]), y));
        });
        auto length_fn = Value::nat([](std::vector<Value>& a) {
            E s = constant(0.0);
            for (auto& v : a) { E x = as_real(v); s = add(s, mul(x, x)); }
            return Value::symbol(fn1(F_SQRT, s));
        });
        m["fast_length"] = length_fn;
        m["length"] = length_fn;
        m["smooth_fmod"] = Value::nat([](std::vector<Value>& a) { return Value::symbol(fn2(F_FMOD, as_real(a.at(0)), as_real(a.at(1)))); });
        m["select"] = Value::nat([](std::vector<Value>& a) { return Value::symbol(select(as_real(a.at(0)), as_real(a.at(1)), as_real(a.at(2)))); });
        for (auto& [name, f] : std::vector<std::pair<std::string, Fn>>{{"lt", F_LT}, {"lte", F_LE}, {"eq", F_EQ}, {"gt", F_GT}, {"gte", F_GE}})
            m[name] = Value::nat([f](std::vector<Value>& a) {
                if (a.at(0).k == V_CPX || a.at(1).k == V_CPX) fail("comparisons are only defined on real values");
                return Value::symbol(fn2(f, as_real(a[0]), as_real(a[1])));
            });
        m["conjugate"] = Value::nat([](std::vector<Value>& a) { return a.at(0).k == V_CPX ? Value::complex(cconj(a[0].c)) : Value::symbol(as_real(a[0])); });
        m["self_conjugate_multiply"] = Value::nat([](std::vector<Value>& a) {
            return a.at(0).k == V_CPX ? Value::symbol(cabs2(a[0].c)) : Value::symbol(mul(as_real(a[0]), as_real(a[0])));
        });
        m["Real"] = Value::nat([](std::vector<Value>& a) { return Value::symbol(as_complex(a.at(0)).re); });
        m["Imaginary"] = Value::nat([](std::vector<Value>& a) { return Value::symbol(as_complex(a.at(0)).im); });
        m["i"] = Value::complex(Cx{constant(0.0), constant(1.0)});
        m["get_i"] = Value::nat([](std::vector<Value>&) { return Value::complex(Cx{constant(0.0), constant(1.0)}); });
        m["M_PI"] = Value::number(M_PI);
        m["PI"] = Value::number(M_PI);
        m["debug"] = Value::nat([](std::vector<Value>&) { return Value(); });
        global->vars["CMath"] = cmath;

        Value math = Value::object();
        (*math.obj)["PI"] = Value::number(M_PI);
        (*math.obj)["E"] = Value::number(M_E);
        auto plain1 = [](double (*f)(double)) { return Value::nat([f](std::vector<Value>& a) { return Value::number(f(plain(a.at(0)))); }); };
        (*math.obj)["sqrt"] = plain1(std::sqrt); (*math.obj)["sin"] = plain1(std::sin); (*math.obj)["cos"] = plain1(std::cos);
        (*math.obj)["abs"] = plain1(std::fabs); (*math.obj)["floor"] = plain1(std::floor);
        (*math.obj)["pow"] = Value::nat([](std::vector<Value>& a) { return Value::number(std::pow(plain(a.at(0)), plain(a.at(1)))); });
        global->vars["Math"] = math;
        global->vars["M_PI"] = Value::number(M_PI);
        global->vars["$pin"] = Value::nat([](std::vector<Value>& a) { return a.empty() ? Value() : a[0]; });
        Value cfg;
        cfg.k = V_CFG;
        global->vars["$cfg"] = cfg;
        Value undef;
        global->vars["undefined"] = undef;
    }

    // ---- evaluation ----

    int call_depth = 0;
    Value call(const Value& f, std::vector<Value>& args) {
        if (f.k == V_NATIVE) return (*f.native)(args);
        if (f.k != V_FUNC) fail("call of a non-function");
        // a script that recurses without end is a script error, not the host's stack (QuickJS: "InternalError: stack overflow"); the
        // deepest chain in the reference's scripts is 6 calls
        struct depth_guard { int& d; depth_guard(int& x) : d(x) { d++; } ~depth_guard() { d--; } } guard(call_depth);
        if (call_depth > 256) fail("script recursion deeper than 256 calls");
        auto env = std::make_shared<Env>();
        env->parent = f.fn->env;
        env->is_function = true;
        const Node& fn = *f.fn->fn;
        for (size_t i = 0; i < fn.params.size(); i++) env->vars[fn.params[i]] = i < args.size() ? args[i] : Value();
        hoist(*fn.kids[0], env);
        try {
            exec_block(*fn.kids[0], env, false);
        } catch (ReturnSignal& r) {
            return r.v;
        }
        return Value();
    }

    // function declarations are visible in their whole scope
    void hoist(const Node& block, const EnvP& env) {
        for (auto& s : block.kids)
            if (s && s->kind == S_FUNCDECL) {
                Value f; f.k = V_FUNC; f.fn = std::make_shared<Closure>(Closure{s->kids[0], env});
                env->vars[s->text] = f;
            }
    }

    void exec_block(const Node& b, const EnvP& env, bool new_scope) {
        EnvP e = env;
        if (new_scope) { e = std::make_shared<Env>(); e->parent = env; hoist(b, e); }
        for (auto& s : b.kids) exec(*s, e);
    }

    void exec(const Node& s, const EnvP& env) {
        switch (s.kind) {
            case S_EMPTY:
            case S_FUNCDECL: return;
            case S_BLOCK: exec_block(s, env, true); return;
            case S_VAR:
                for (auto& [name, init] : s.decls) {
                    Env* scope = s.block_scoped ? env.get() : env->function_scope();
                    Value v = init ? eval(*init, env) : Value();
                    if (init || !scope->vars.count(name)) scope->vars[name] = v;
                }
                return;
            case S_EXPR: completion = eval(*s.kids[0], env); return;
            case S_RETURN: throw ReturnSignal{s.kids.empty() ? Value() : eval(*s.kids[0], env)};
            case S_IF:
                if (truthy(eval(*s.kids[0], env))) exec(*s.kids[1], env);
                else if (s.kids.size() > 2) exec(*s.kids[2], env);
                return;
            case S_FOR: {
                auto scope = std::make_shared<Env>();
                scope->parent = env;
                if (s.kids[0]) exec(*s.kids[0], scope);
                for (int guard = 0; !s.kids[1] || truthy(eval(*s.kids[1], scope)); guard++) {
                    if (guard > 1000000) fail("script loop does not terminate");
                    exec(*s.kids[3], scope);
                    if (s.kids[2]) eval(*s.kids[2], scope);
                }
                return;
            }
            case S_WHILE:
                for (int guard = 0; truthy(eval(*s.kids[0], env)); guard++) {
                    if (guard > 1000000) fail("script loop does not terminate");
                    exec(*s.kids[1], env);
                }
                return;
            default: fail("bad statement");
        }
    }

    Value get_member(const Value& o, const std::string& prop) {
        if (o.k == V_CFG) {
            // js_interop.cpp:795-815: first mention registers the parameter; the value is the symbol cfg->NAME
            for (char ch : prop) if (!(isalnum((unsigned char)ch) || ch == '_')) fail("Value must be alphanumeric or _");
            vars->add(prop, 0.f);
            Value v = Value::symbol(var("cfg->" + prop));
            v.str = prop;
            return v;
        }
        if (o.k == V_OBJ) {
            auto it = o.obj->find(prop);
            return it == o.obj->end() ? Value() : it->second;
        }
        if (o.k == V_ARR && prop == "length") return Value::number((double)o.arr->size());
        if (o.k == V_STR && prop == "length") return Value::number((double)o.str.size());
        if (o.k == V_SYM && prop == "$default") return Value();
        fail("cannot read property '" + prop + "'");
    }

    void assign_to(const Node& target, const Value& v, const EnvP& env) {
        if (target.kind == N_IDENT) {
            Value* slot = env->find(target.text);
            if (slot) *slot = v;
            else global->vars[target.text] = v;   // sloppy-mode implicit global
            return;
        }
        if (target.kind == N_MEMBER) {
            Value o = eval(*target.kids[0], env);
            if (o.k == V_SYM && target.text == "$default") {
                // js_interop.cpp:762-793
                if (o.str.empty()) fail("Must be pseudoconstant value in $default set");
                if (!is_plain(v)) fail("$default must be a number");
                vars->set_default(o.str, (float)plain(v));
                return;
            }
            if (o.k == V_ARR && target.text == "length") {
                if (!is_plain(v)) fail("bad array length");
                o.arr->resize((size_t)plain(v));
                return;
            }
            if (o.k == V_OBJ) { (*o.obj)[target.text] = v; return; }
            if (o.k == V_CFG) return;   // "Warning, setting a config from js" (js_interop.cpp:817-824)
            fail("cannot assign property '" + target.text + "'");
        }
        if (target.kind == N_INDEX) {
            Value o = eval(*target.kids[0], env);
            Value idx = eval(*target.kids[1], env);
            if (o.k == V_ARR) {
                if (!is_plain(idx)) fail("array index must be a plain number");
                double d = plain(idx);
                if (d < 0 || d != std::floor(d) || d > 1e6) fail("bad array index");
                if ((size_t)d >= o.arr->size()) o.arr->resize((size_t)d + 1);
                (*o.arr)[(size_t)d] = v;
                return;
            }
            if (o.k == V_OBJ && idx.k == V_STR) { (*o.obj)[idx.str] = v; return; }
            fail("cannot index-assign this value");
        }
        fail("bad assignment target");
    }

    Value eval(const Node& n, const EnvP& env) {
        switch (n.kind) {
            case N_NUM: return Value::number(n.num);
            case N_STR: { Value s; s.k = V_STR; s.str = n.text; return s; }
            case N_IDENT: {
                if (n.text == "true") return Value::boolean(true);
                if (n.text == "false") return Value::boolean(false);
                if (n.text == "null") return Value();
                Value* v = env->find(n.text);
                if (!v) fail("script line " + std::to_string(n.line) + ": '" + n.text + "' is not defined");
                return *v;
            }
            case N_ARRAY: {
                Value a = Value::array();
                for (auto& k : n.kids) a.arr->push_back(eval(*k, env));
                return a;
            }
            case N_OBJECT: {
                Value o = Value::object();
                for (auto& [k, v] : n.decls) (*o.obj)[k] = eval(*v, env);
                return o;
            }
            case N_FUNC: {
                Value f; f.k = V_FUNC; f.fn = std::make_shared<Closure>(Closure{std::make_shared<Node>(n), env});
                return f;
            }
            case N_MEMBER: return get_member(eval(*n.kids[0], env), n.text);
            case N_INDEX: {
                Value o = eval(*n.kids[0], env);
                Value idx = eval(*n.kids[1], env);
                if (o.k == V_ARR) {
                    if (!is_plain(idx)) fail("array index must be a plain number");
                    double d = plain(idx);
                    if (d < 0 || d != std::floor(d) || (size_t)d >= o.arr->size()) return Value();
                    return (*o.arr)[(size_t)d];
                }
                if (o.k == V_OBJ && idx.k == V_STR) return get_member(o, idx.str);
                fail("cannot index this value");
            }
            case N_CALL: {
                Value f = eval(*n.kids[0], env);
                std::vector<Value> args;
                for (size_t i = 1; i < n.kids.size(); i++) args.push_back(eval(*n.kids[i], env));
                return call(f, args);
            }
            case N_UNARY: {
                Value a = eval(*n.kids[0], env);
                if (n.text == "!") return Value::boolean(!truthy(a));
                if (n.text == "+") return a;
                if (is_plain(a)) return Value::number(-plain(a));
                if (a.k == V_CPX) return Value::complex(cneg(a.c));
                return Value::symbol(neg(as_real(a)));
            }
            case N_UPDATE: {
                Value old = eval(*n.kids[0], env);
                if (!is_plain(old)) fail("++/-- on a non-number");
                Value nv = Value::number(plain(old) + (n.text == "++" ? 1 : -1));
                assign_to(*n.kids[0], nv, env);
                return n.prefix ? nv : Value::number(plain(old));
            }
            case N_BINARY: {
                const std::string& op = n.text;
                if (op == "&&") { Value a = eval(*n.kids[0], env); return truthy(a) ? eval(*n.kids[1], env) : a; }
                if (op == "||") { Value a = eval(*n.kids[0], env); return truthy(a) ? a : eval(*n.kids[1], env); }
                Value a = eval(*n.kids[0], env), b = eval(*n.kids[1], env);
                if (op == "==" || op == "===" || op == "!=" || op == "!==" || op == "<" || op == ">" || op == "<=" || op == ">=") {
                    bool r;
                    if (a.k == V_STR && b.k == V_STR) {
                        r = op[0] == '=' ? a.str == b.str : op[0] == '!' ? a.str != b.str : op == "<" ? a.str < b.str : op == ">" ? a.str > b.str : op == "<=" ? a.str <= b.str : a.str >= b.str;
                    } else {
                        auto num = [](const Value& v) -> double {
                            if (is_plain(v)) return plain(v);
                            if (v.k == V_SYM && is_const(v.e)) return v.e->c;
                            if (v.k == V_UNDEF) return NAN;
                            fail("comparison operators work on plain numbers only (use CMath.lt / lte / eq / gt / gte on symbolic values)");
                        };
                        double x = num(a), y = num(b);
                        r = op[0] == '=' ? x == y : op[0] == '!' ? x != y : op == "<" ? x < y : op == ">" ? x > y : op == "<=" ? x <= y : x >= y;
                    }
                    return Value::boolean(r);
                }
                return arith(op, a, b);
            }
            case N_ASSIGN: {
                Value v = eval(*n.kids[1], env);
                if (n.text != "=") v = arith(n.text.substr(0, 1), eval(*n.kids[0], env), v);
                assign_to(*n.kids[0], v, env);
                return v;
            }
            case N_COND: return truthy(eval(*n.kids[0], env)) ? eval(*n.kids[1], env) : eval(*n.kids[2], env);
            default: fail("bad expression");
        }
    }

    // evaluates a script; its completion value must be a function (js_interop.cpp:848-853)
    Value load(const std::string& source, const std::string& what) {
        Parser ps(source);
        NodeP prog = ps.program();
        hoist(*prog, global);
        completion = Value();
        try {
            exec_block(*prog, global, false);
        } catch (ReturnSignal&) {
            fail(what + ": return outside of a function");
        }
        if (completion.k != V_FUNC && completion.k != V_NATIVE) fail("Expected function in eval of script " + what);
        return completion;
    }
};

// ------------------------------------------------------------------------------------------------
// files

std::string read_text(const std::string& path) {
    std::ifstream f(path, std::ios::binary);
    if (!f) fail("cannot read " + path);
    std::stringstream ss;
    ss << f.rdbuf();
    return ss.str();
}

bool exists(const std::string& path) {
    struct stat st;
    return stat(path.c_str(), &st) == 0;
}

// flat JSON object -> key/value strings (strings unescaped, numbers / true / false verbatim)
std::map<std::string, std::string> parse_flat_json(const std::string& text, const std::string& what) {
    std::map<std::string, std::string> out;
    size_t i = 0, n = text.size();
    auto ws = [&]() { while (i < n && isspace((unsigned char)text[i])) i++; };
    auto str = [&]() {
        std::string s;
        if (text[i] != '"') fail(what + ": expected string");
        i++;
        while (i < n && text[i] != '"') {
            if (text[i] == '\\' && i + 1 < n) {
                i++;
                char c = text[i];
                s += c == 'n' ? '\n' : c == 't' ? '\t' : c == 'r' ? '\r' : c;
            } else {
                s += text[i];
            }
            i++;
        }
        i++;
        return s;
    };
    ws();
    if (i >= n || text[i] != '{') fail(what + ": expected a JSON object");
    i++;
    for (;;) {
        ws();
        if (i < n && text[i] == '}') break;
        std::string key = str();
        ws();
        if (i >= n || text[i] != ':') fail(what + ": expected ':'");
        i++;
        ws();
        std::string value;
        if (text[i] == '"') value = str();
        else {
            size_t j = i;
            while (j < n && text[j] != ',' && text[j] != '}' && !isspace((unsigned char)text[j])) j++;
            value = text.substr(i, j - i);
            i = j;
        }
        out[key] = value;
        ws();
        if (i < n && text[i] == ',') { i++; continue; }
        ws();
        if (i < n && text[i] == '}') break;
        fail(what + ": malformed JSON");
    }
    return out;
}

std::string find_script(const std::string& dir, const std::string& name) {
    for (const char* sub : {"", "coordinates/", "origins/"}) {
        std::string p = dir + "/" + sub + name + ".js";
        if (exists(p)) return p;
    }
    fail("Could not lookup " + name);
}

struct ScriptSet {
    std::vector<std::shared_ptr<Interpreter>> interpreters;
};

std::vector<E> to_exprs(const Value& v, const std::string& what) {
    if (v.k != V_ARR) fail(what + ": Must return array");
    std::vector<E> out;
    for (auto& x : *v.arr) out.push_back(as_real(x));
    return out;
}

}  // namespace

std::shared_ptr<void> load_metric_from_scripts(const std::string& dir, const std::string& name, MetricConfig& cfg, MetricFunctions& f,
                                               DynamicVars& vars) {
    // content_manager.cpp:70-112: own JSON on top of one level of inherit_settings
    std::string json_path = dir + "/" + name + ".json";
    if (!exists(json_path)) fail("no config " + json_path);
    auto own = parse_flat_json(read_text(json_path), json_path);
    cfg = MetricConfig();
    auto inherit = own.find("inherit_settings");
    if (inherit != own.end()) {
        std::string parent_path = dir + "/" + inherit->second + ".json";
        if (exists(parent_path)) cfg.apply(parse_flat_json(read_text(parent_path), parent_path));
    }
    cfg.apply(own);

    auto set = std::make_shared<ScriptSet>();
    vars = DynamicVars();
    DynamicVars* shared_vars = &vars;   // one sandbox: every script of the metric sees the same $cfg (js_interop.hpp:25-30)

    auto load_fn = [&](const std::string& path) {
        auto in = std::make_shared<Interpreter>(shared_vars);
        Value fn = in->load(read_text(path), path);
        set->interpreters.push_back(in);
        return std::make_pair(in, fn);
    };
    auto make_fn4 = [&](const std::string& path, const std::string& what) -> Fn4 {
        auto [in, fn] = load_fn(path);
        auto interp = in;
        Value func = fn;
        return [interp, func, what](E a, E b, E c, E d) {
            std::vector<Value> args = {Value::symbol(a), Value::symbol(b), Value::symbol(c), Value::symbol(d)};
            return to_exprs(interp->call(func, args), what);
        };
    };

    std::string metric_path = dir + "/" + name + ".js";
    if (!exists(metric_path)) fail("No .js file for metric " + name);
    f.metric = make_fn4(metric_path, metric_path);
    f.to_polar = make_fn4(find_script(dir, cfg.to_polar), cfg.to_polar);
    f.from_polar = make_fn4(find_script(dir, cfg.from_polar), cfg.from_polar);
    {
        auto [in, fn] = load_fn(find_script(dir, cfg.origin_distance));
        auto interp = in;
        Value func = fn;
        f.origin_distance = [interp, func](E a, E b, E c, E d) {
            std::vector<Value> args = {Value::symbol(a), Value::symbol(b), Value::symbol(c), Value::symbol(d)};
            return as_real(interp->call(func, args));
        };
    }
    if (!cfg.coordinate_periodicity.empty()) f.coordinate_periodicity = make_fn4(find_script(dir, cfg.coordinate_periodicity), cfg.coordinate_periodicity);
    else f.coordinate_periodicity = nullptr;
    return set;
}

}  // namespace gr
