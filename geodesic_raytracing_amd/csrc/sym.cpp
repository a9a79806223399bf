// sym.cpp — see sym.hpp.
#include "sym.hpp"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <deque>
#include <functional>
#include <mutex>
#include <stdexcept>
#include <unordered_set>

namespace sym {

namespace {

struct Key {
    Op op;
    Fn fn;
    uint64_t cbits;
    std::string name;
    uint32_t a, b, s;
    bool operator==(const Key& o) const {
        return op == o.op && fn == o.fn && cbits == o.cbits && a == o.a && b == o.b && s == o.s &&
               name == o.name;
    }
};
struct KeyHash {
    size_t operator()(const Key& k) const {
        uint64_t h = 1469598103934665603ull;
        auto mix = [&](uint64_t v) {
            h ^= v + 0x9e3779b97f4a7c15ull + (h << 6) + (h >> 2);
        };
        mix(k.op);
        mix(k.fn);
        mix(k.cbits);
        mix(std::hash<std::string>()(k.name));
        mix(k.a);
        mix(k.b);
        mix(k.s);
        return (size_t)h;
    }
};

struct Pool {
    std::deque<Node> nodes;
    std::unordered_map<Key, E, KeyHash> table;
    std::recursive_mutex mu;
};
Pool& pool() {
    static Pool p;
    return p;
}

uint32_t var_deps(const std::string& n) {
    if (n == "v1") return DEP_V1;
    if (n == "v2") return DEP_V2;
    if (n == "v3") return DEP_V3;
    if (n == "v4") return DEP_V4;
    if (n.size() == 3 && n[0] == 'i' && n[1] == 'v') return DEP_IV;
    if (n.size() == 3 && n[0] == 'd' && n[1] == 'v') return DEP_DV;
    if (n.rfind("cfg->", 0) == 0) return DEP_CFG;
    return DEP_OTHER;
}

E intern(Op op, Fn fn, double c, const std::string& name, E a, E b, E s) {
    Pool& p = pool();
    std::lock_guard<std::recursive_mutex> lock(p.mu);
    Key k;
    k.op = op;
    k.fn = fn;
    uint64_t bits = 0;
    if (op == CONST) {
        if (c == 0.0) c = 0.0;  // collapse -0.0
        std::memcpy(&bits, &c, sizeof(bits));
    }
    k.cbits = bits;
    k.name = name;
    k.a = a ? a->id + 1 : 0;
    k.b = b ? b->id + 1 : 0;
    k.s = s ? s->id + 1 : 0;
    auto it = p.table.find(k);
    if (it != p.table.end()) return it->second;
    Node n;
    n.op = op;
    n.fn = fn;
    n.c = c;
    n.name = name;
    n.a = a;
    n.b = b;
    n.s = s;
    n.id = (uint32_t)p.nodes.size();
    n.deps = 0;
    uint64_t size = 1;
    if (op == VAR) n.deps = var_deps(name);
    if (a) { n.deps |= a->deps; size += a->size; }
    if (b) { n.deps |= b->deps; size += b->size; }
    if (s) { n.deps |= s->deps; size += s->size; }
    n.size = size > 0x7fffffffu ? 0x7fffffffu : (uint32_t)size;
    {
        auto mix = [](uint64_t h, uint64_t v) {
            h ^= v + 0x9e3779b97f4a7c15ull + (h << 6) + (h >> 2);
            h *= 0xff51afd7ed558ccdull;
            return h ^ (h >> 33);
        };
        uint64_t h = mix(0x243f6a8885a308d3ull, (uint64_t)op * 131 + (uint64_t)fn);
        h = mix(h, bits);
        for (unsigned char ch : name) h = mix(h, ch);
        h = mix(h, a ? a->shape : 1);
        h = mix(h, b ? b->shape : 2);
        h = mix(h, s ? s->shape : 3);
        n.shape = h;
    }
    p.nodes.push_back(n);
    E e = &p.nodes.back();
    p.table.emplace(k, e);
    return e;
}

// canonical order of the operands of a commutative node
inline bool ordered_before(E x, E y) { return x->shape != y->shape ? x->shape < y->shape : x->id < y->id; }

double apply1(Fn f, double x) {
    switch (f) {
        case F_SIN: return std::sin(x);
        case F_COS: return std::cos(x);
        case F_TAN: return std::tan(x);
        case F_ASIN: return std::asin(x);
        case F_ACOS: return std::acos(x);
        case F_ATAN: return std::atan(x);
        case F_EXP: return std::exp(x);
        case F_LOG: return std::log(x);
        case F_SQRT: return std::sqrt(x);
        case F_FABS: return std::fabs(x);
        case F_SINH: return std::sinh(x);
        case F_COSH: return std::cosh(x);
        case F_TANH: return std::tanh(x);
        case F_SIGN: return x > 0 ? 1.0 : (x < 0 ? -1.0 : 0.0);
        case F_EXP2_FAST: return std::exp2(x);
        case F_RSQRT_FAST: return 1.0 / std::sqrt(x);
        case F_SIN2: return std::sin(x) * std::sin(x);
        case F_COS2: return std::cos(x) * std::cos(x);
        case F_SINCOS: return std::sin(x) * std::cos(x);
        default: throw std::runtime_error("apply1: bad function");
    }
}

double apply2(Fn f, double x, double y) {
    switch (f) {
        case F_ATAN2: return std::atan2(x, y);
        case F_POW: return std::pow(x, y);
        case F_FMOD: return std::fmod(x, y);
        case F_MIN: return std::min(x, y);
        case F_MAX: return std::max(x, y);
        case F_CSQRT_RE: return std::sqrt(std::max(0.5 * (std::hypot(x, y) + x), 0.0));
        case F_CSQRT_IM: return (y < 0 ? -1.0 : 1.0) * std::sqrt(std::max(0.5 * (std::hypot(x, y) - x), 0.0));
        case F_LT: return x < y ? 1.0 : 0.0;
        case F_LE: return x <= y ? 1.0 : 0.0;
        case F_EQ: return x == y ? 1.0 : 0.0;
        case F_GT: return x > y ? 1.0 : 0.0;
        case F_GE: return x >= y ? 1.0 : 0.0;
        default: throw std::runtime_error("apply2: bad function");
    }
}

const char* fn_name(Fn f) {
    switch (f) {
        case F_SIN: return "sin";
        case F_COS: return "cos";
        case F_TAN: return "tan";
        case F_ASIN: return "asin";
        case F_ACOS: return "acos";
        case F_ATAN: return "atan";
        case F_EXP: return "exp";
        case F_LOG: return "log";
        case F_SQRT: return "sqrt";
        case F_FABS: return "fabs";
        case F_SINH: return "sinh";
        case F_COSH: return "cosh";
        case F_TANH: return "tanh";
        case F_SIGN: return "sign";
        case F_EXP2_FAST: return "gr_exp2";
        case F_RSQRT_FAST: return "gr_rsqrt";
        case F_SIN2: return "gr_sin2";
        case F_COS2: return "gr_cos2";
        case F_SINCOS: return "gr_sincos";
        case F_ATAN2: return "atan2";
        case F_POW: return "pow";
        case F_FMOD: return "fmod";
        case F_MIN: return "fmin";
        case F_MAX: return "fmax";
        default: return "?";
    }
}

bool is_cmp(Fn f) { return f == F_LT || f == F_LE || f == F_EQ || f == F_GT || f == F_GE; }

// float-valued constant folding: every literal is rounded to float when printed, and the kernels
// compute in fp32, so fold through float to keep folded and unfolded evaluation consistent.
double fl(double v) { return (double)(float)v; }

}  // namespace

namespace {
// e == coef * term with a literal coefficient
void split_coef(E e, double& coef, E& term) {
    if (e->op == MUL && e->a->op == CONST) { coef = e->a->c; term = e->b; }
    else if (e->op == NEG) { split_coef(e->a, coef, term); coef = -coef; }
    else { coef = 1.0; term = e; }
}
}  // namespace

E constant(double v) { return intern(CONST, F_NONE, v, "", nullptr, nullptr, nullptr); }
E var(const std::string& name) { return intern(VAR, F_NONE, 0, name, nullptr, nullptr, nullptr); }

E neg(E a) {
    if (a->op == CONST) return constant(-a->c);
    if (a->op == NEG) return a->a;
    if (a->op == SUB) return sub(a->b, a->a);
    return intern(NEG, F_NONE, 0, "", a, nullptr, nullptr);
}

E add(E a, E b) {
    if (a->op == CONST && b->op == CONST) return constant(a->c + b->c);
    if (is_zero(a)) return b;
    if (is_zero(b)) return a;
    if (b->op == NEG) return sub(a, b->a);
    if (a->op == NEG) return sub(b, a->a);
    if (b->op == CONST && b->c < 0) return sub(a, constant(-b->c));
    // a term and its opposite one level down cancel: (x - y) + y = x  (round 5, with the rules in sub: the inverse of a warp drive's
    // t-x block has the determinant (v^2 f^2 - 1) - v^2 f^2, which is -1 and was evaluated - and divided by - every attempt)
    if (a->op == SUB && a->b == b) return a->a;
    if (b->op == SUB && b->b == a) return b->a;
    {
        double ca, cb; E ta, tb;
        split_coef(a, ca, ta);
        split_coef(b, cb, tb);
        if (ta == tb && ta->op != CONST) return mul(constant(ca + cb), ta);
    }
    if (ordered_before(b, a)) std::swap(a, b);
    return intern(ADD, F_NONE, 0, "", a, b, nullptr);
}

E sub(E a, E b) {
    if (a->op == CONST && b->op == CONST) return constant(a->c - b->c);
    if (is_zero(b)) return a;
    if (is_zero(a)) return neg(b);
    if (a == b) return constant(0.0);
    if (b->op == NEG) return add(a, b->a);
    if (a->op == NEG) return neg(add(a->a, b));
    if (b->op == CONST && b->c < 0) return add(a, constant(-b->c));
    if (a->op == SUB && a->a == b) return neg(a->b);            // (x - y) - x = -y
    if (a->op == ADD && a->a == b) return a->b;                 // (x + y) - x = y
    if (a->op == ADD && a->b == b) return a->a;                 // (y + x) - x = y
    if (b->op == ADD && b->a == a) return neg(b->b);            // x - (x + y) = -y
    if (b->op == ADD && b->b == a) return neg(b->a);            // x - (y + x) = -y
    if (b->op == SUB && b->a == a) return b->b;                 // x - (x - y) = y
    {
        double ca, cb; E ta, tb;
        split_coef(a, ca, ta);
        split_coef(b, cb, tb);
        if (ta == tb && ta->op != CONST) return mul(constant(ca - cb), ta);
    }
    return intern(SUB, F_NONE, 0, "", a, b, nullptr);
}

E mul(E a, E b) {
    if (a->op == CONST && b->op == CONST) return constant(a->c * b->c);
    if (is_zero(a) || is_zero(b)) return constant(0.0);
    if (is_one(a)) return b;
    if (is_one(b)) return a;
    if (b->op == CONST) std::swap(a, b);  // constants on the left
    if (a->op == CONST) {
        if (a->c == -1.0) return neg(b);
        if (a->c < 0) return neg(mul(constant(-a->c), b));
        if (b->op == MUL && b->a->op == CONST) return mul(constant(a->c * b->a->c), b->b);
        if (b->op == NEG) return neg(mul(a, b->a));
        if (b->op == DIV && b->a->op == CONST) return div(constant(a->c * b->a->c), b->b);
        return intern(MUL, F_NONE, 0, "", a, b, nullptr);
    }
    if (a->op == NEG && b->op == NEG) return mul(a->a, b->a);
    if (a->op == NEG) return neg(mul(a->a, b));
    if (b->op == NEG) return neg(mul(a, b->a));
    // float constants migrate outwards: (c*x)*y -> c*(x*y)
    if (a->op == MUL && a->a->op == CONST) return mul(a->a, mul(a->b, b));
    if (b->op == MUL && b->a->op == CONST) return mul(b->a, mul(a, b->b));
    if (ordered_before(b, a)) std::swap(a, b);
    return intern(MUL, F_NONE, 0, "", a, b, nullptr);
}

E div(E a, E b) {
    if (b->op == CONST && b->c == 0.0) {
        // keep the division visible (the scripts never do this on purpose)
        return intern(DIV, F_NONE, 0, "", a, b, nullptr);
    }
    if (a->op == CONST && b->op == CONST) return constant(a->c / b->c);
    if (is_zero(a)) return constant(0.0);
    if (is_one(b)) return a;
    if (b->op == CONST) return mul(constant(1.0 / b->c), a);
    if (a == b) return constant(1.0);
    if (a->op == NEG && b->op == NEG) return div(a->a, b->a);
    if (a->op == NEG) return neg(div(a->a, b));
    if (b->op == NEG) return neg(div(a, b->a));
    if (a->op == CONST && a->c < 0) return neg(div(constant(-a->c), b));
    if (b->op == MUL && b->a->op == CONST && a->op == CONST) return div(constant(a->c / b->a->c), b->b);
    if (b->op == MUL && b->a->op == CONST) return mul(constant(1.0 / b->a->c), div(a, b->b));
    if (a->op == MUL && a->a->op == CONST) return mul(a->a, div(a->b, b));
    if (b->op == DIV) return div(mul(a, b->b), b->a);   // a/(p/q) -> (a*q)/p
    return intern(DIV, F_NONE, 0, "", a, b, nullptr);
}

E powi(E a, int n) {
    if (n == 0) return constant(1.0);
    if (n < 0) return div(constant(1.0), powi(a, -n));
    E result = nullptr;
    E base = a;
    while (n) {
        if (n & 1) result = result ? mul(result, base) : base;
        n >>= 1;
        if (n) base = mul(base, base);
    }
    return result;
}

E fn1(Fn f, E a) {
    if (a->op == CONST) {
        double r = apply1(f, fl(a->c));
        if (std::isfinite(r)) return constant(fl(r));
    }
    switch (f) {
        case F_SIN:
        case F_TAN:
        case F_ASIN:
        case F_ATAN:
        case F_SINH:
        case F_TANH:
        case F_SIGN:
            if (a->op == NEG) return neg(fn1(f, a->a));  // odd functions
            break;
        case F_COS:
        case F_COSH:
        case F_FABS:
            if (a->op == NEG) return fn1(f, a->a);  // even functions
            break;
        default: break;
    }
    if (f == F_FABS && a->op == FN1 && (a->fn == F_FABS || a->fn == F_SQRT || a->fn == F_EXP || a->fn == F_COSH))
        return a;
    if (f == F_FABS && a->op == MUL && a->a == a->b) return a;  // |x*x|
    return intern(FN1, f, 0, "", a, nullptr, nullptr);
}

// e >= 0 for every real value of its variables, as far as the structure shows it (sums of squares, roots, moduli ...)
static bool provably_nonnegative(E e, int depth = 0) {
    if (depth > 64) return false;
    switch (e->op) {
        case CONST: return e->c >= 0.0;
        case MUL: return e->a == e->b || (provably_nonnegative(e->a, depth + 1) && provably_nonnegative(e->b, depth + 1));
        case ADD:
        case DIV: return provably_nonnegative(e->a, depth + 1) && provably_nonnegative(e->b, depth + 1);
        case FN1: return e->fn == F_SQRT || e->fn == F_FABS || e->fn == F_EXP || e->fn == F_COSH;
        case FN2:
            if (e->fn == F_MAX) return provably_nonnegative(e->a, depth + 1) || provably_nonnegative(e->b, depth + 1);
            if (e->fn == F_MIN) return provably_nonnegative(e->a, depth + 1) && provably_nonnegative(e->b, depth + 1);
            return e->fn == F_CSQRT_RE;
        default: return false;
    }
}

E fn2(Fn f, E a, E b) {
    if (a->op == CONST && b->op == CONST) {
        double r = apply2(f, fl(a->c), fl(b->c));
        if (std::isfinite(r)) return constant(fl(r));
    }
    // principal root of a complex number that turns out to be real (a script written for complex parameters, substituted with
    // values that make them real - the rod half-lengths of double Kerr with sub-extreme constituents): sqrt(a) and 0 on the
    // non-negative axis, sqrt(max(a, 0)) + i sqrt(max(-a, 0)) where the sign of a is not known
    if ((f == F_CSQRT_RE || f == F_CSQRT_IM) && is_zero(b)) {
        if (provably_nonnegative(a)) return f == F_CSQRT_RE ? fn1(F_SQRT, a) : constant(0.0);
        return fn1(F_SQRT, fn2(F_MAX, f == F_CSQRT_RE ? a : neg(a), constant(0.0)));
    }
    if (f == F_POW && b->op == CONST) {
        double n = b->c;
        if (n == std::floor(n) && std::fabs(n) <= 8) return powi(a, (int)n);
        if (n == 0.5) return fn1(F_SQRT, a);
        if (n == -0.5) return div(constant(1.0), fn1(F_SQRT, a));
    }
    return intern(FN2, f, 0, "", a, b, nullptr);
}

E select(E cond, E t, E f) {
    if (cond->op == CONST) return cond->c != 0.0 ? t : f;
    if (t == f) return t;
    return intern(SELECT, F_NONE, 0, "", cond, t, f);
}

// ----------------------------------------------------------------------------------------------

namespace {
struct DiffKey {
    uint32_t id;
    std::string v;
    bool operator==(const DiffKey& o) const { return id == o.id && v == o.v; }
};
struct DiffKeyHash {
    size_t operator()(const DiffKey& k) const { return std::hash<std::string>()(k.v) * 1000003u + k.id; }
};
std::unordered_map<DiffKey, E, DiffKeyHash>& diff_memo() {
    static std::unordered_map<DiffKey, E, DiffKeyHash> m;
    return m;
}
}  // namespace

E diff(E e, const std::string& wrt) {
    if (e->op == CONST) return constant(0.0);
    if (e->op == VAR) return constant(e->name == wrt ? 1.0 : 0.0);
    if ((e->deps & var_deps(wrt)) == 0) return constant(0.0);
    std::lock_guard<std::recursive_mutex> lock(pool().mu);
    DiffKey key{e->id, wrt};
    auto& memo = diff_memo();
    auto it = memo.find(key);
    if (it != memo.end()) return it->second;
    E r = nullptr;
    E a = e->a, b = e->b;
    switch (e->op) {
        case ADD: r = add(diff(a, wrt), diff(b, wrt)); break;
        case SUB: r = sub(diff(a, wrt), diff(b, wrt)); break;
        case NEG: r = neg(diff(a, wrt)); break;
        case MUL: r = add(mul(diff(a, wrt), b), mul(a, diff(b, wrt))); break;
        case DIV: {
            E da = diff(a, wrt), db = diff(b, wrt);
            if (is_zero(db)) r = div(da, b);
            else if (is_zero(da)) r = neg(div(mul(a, db), mul(b, b)));
            else r = div(sub(mul(da, b), mul(a, db)), mul(b, b));
            break;
        }
        case FN1: {
            E da = diff(a, wrt);
            if (is_zero(da)) { r = da; break; }
            switch (e->fn) {
                case F_SIN: r = mul(fn1(F_COS, a), da); break;
                case F_COS: r = neg(mul(fn1(F_SIN, a), da)); break;
                case F_TAN: r = mul(add(constant(1.0), mul(e, e)), da); break;
                case F_ASIN: r = div(da, fn1(F_SQRT, sub(constant(1.0), mul(a, a)))); break;
                case F_ACOS: r = neg(div(da, fn1(F_SQRT, sub(constant(1.0), mul(a, a))))); break;
                case F_ATAN: r = div(da, add(constant(1.0), mul(a, a))); break;
                case F_EXP: r = mul(e, da); break;
                case F_LOG: r = div(da, a); break;
                case F_SQRT: r = div(mul(constant(0.5), da), e); break;
                case F_FABS: r = mul(fn1(F_SIGN, a), da); break;
                case F_SINH: r = mul(fn1(F_COSH, a), da); break;
                case F_COSH: r = mul(fn1(F_SINH, a), da); break;
                case F_TANH: r = mul(sub(constant(1.0), mul(e, e)), da); break;
                case F_SIGN: r = constant(0.0); break;
                default: throw std::runtime_error("diff: bad unary function");
            }
            break;
        }
        case FN2: {
            E da = diff(a, wrt), db = diff(b, wrt);
            switch (e->fn) {
                case F_ATAN2:  // atan2(a=y, b=x)
                    r = div(sub(mul(b, da), mul(a, db)), add(mul(a, a), mul(b, b)));
                    break;
                case F_POW:
                    if (is_zero(db)) {
                        r = mul(mul(b, fn2(F_POW, a, sub(b, constant(1.0)))), da);
                    } else {
                        r = mul(e, add(mul(db, fn1(F_LOG, a)), div(mul(b, da), a)));
                    }
                    break;
                case F_FMOD: r = da; break;
                case F_CSQRT_RE:
                case F_CSQRT_IM: {
                    // w = sqrt(a + i b):  dw = (da + i db) conj(w) / (2 |w|^2),  |w|^2 = |a + i b|
                    E wr = fn2(F_CSQRT_RE, a, b), wi = fn2(F_CSQRT_IM, a, b);
                    E den = mul(constant(2.0), fn1(F_SQRT, add(mul(a, a), mul(b, b))));
                    if (e->fn == F_CSQRT_RE) r = div(add(mul(da, wr), mul(db, wi)), den);
                    else r = div(sub(mul(db, wr), mul(da, wi)), den);
                    break;
                }
                case F_MIN: r = select(fn2(F_LT, a, b), da, db); break;
                case F_MAX: r = select(fn2(F_GT, a, b), da, db); break;
                default: r = constant(0.0); break;  // comparisons
            }
            break;
        }
        case SELECT: r = select(e->a, diff(e->b, wrt), diff(e->s, wrt)); break;
        default: throw std::runtime_error("diff: bad op");
    }
    memo.emplace(key, r);
    return r;
}

namespace {
E subst_rec(E e, const std::map<std::string, E>& m, std::unordered_map<E, E>& memo) {
    if (e->op == CONST) return e;
    if (e->op == VAR) {
        auto it = m.find(e->name);
        return it == m.end() ? e : it->second;
    }
    auto it = memo.find(e);
    if (it != memo.end()) return it->second;
    E r = nullptr;
    switch (e->op) {
        case ADD: r = add(subst_rec(e->a, m, memo), subst_rec(e->b, m, memo)); break;
        case SUB: r = sub(subst_rec(e->a, m, memo), subst_rec(e->b, m, memo)); break;
        case MUL: r = mul(subst_rec(e->a, m, memo), subst_rec(e->b, m, memo)); break;
        case DIV: r = div(subst_rec(e->a, m, memo), subst_rec(e->b, m, memo)); break;
        case NEG: r = neg(subst_rec(e->a, m, memo)); break;
        case FN1: r = fn1(e->fn, subst_rec(e->a, m, memo)); break;
        case FN2: r = fn2(e->fn, subst_rec(e->a, m, memo), subst_rec(e->b, m, memo)); break;
        case SELECT:
            r = select(subst_rec(e->a, m, memo), subst_rec(e->b, m, memo), subst_rec(e->s, m, memo));
            break;
        default: throw std::runtime_error("subst: bad op");
    }
    memo.emplace(e, r);
    return r;
}
}  // namespace

E share_reciprocals(E e, std::unordered_map<E, E>& memo) {
    if (e->op == CONST || e->op == VAR) return e;
    auto it = memo.find(e);
    if (it != memo.end()) return it->second;
    auto R = [&](E x) { return share_reciprocals(x, memo); };
    E r = nullptr;
    switch (e->op) {
        case ADD: r = add(R(e->a), R(e->b)); break;
        case SUB: r = sub(R(e->a), R(e->b)); break;
        case MUL: r = mul(R(e->a), R(e->b)); break;
        case DIV: {
            E n = R(e->a), d = R(e->b);
            if (d->op == MUL && d->a == d->b) {
                E inv = div(constant(1.0), d->a);
                r = mul(n, mul(inv, inv));
            } else {
                r = div(n, d);
            }
            break;
        }
        case NEG: r = neg(R(e->a)); break;
        case FN1: r = fn1(e->fn, R(e->a)); break;
        case FN2: r = fn2(e->fn, R(e->a), R(e->b)); break;
        case SELECT: r = select(R(e->a), R(e->b), R(e->s)); break;
        default: throw std::runtime_error("share_reciprocals: bad op");
    }
    memo.emplace(e, r);
    return r;
}

// ---- coordinates there and back again ------------------------------------------------------------------------------------------
// Rewrites, bottom-up, the patterns that `from_polar(to_polar(x))` leaves behind: sin / cos of an atan2 become ratios over the
// hypotenuse, sqrt(A) * sqrt(A) is A, x * (y / x) is y, x / x is 1.  NOT an identity at the points where a hypotenuse vanishes
// (atan2(0, 0) is 0 by convention, the ratio 0 / 0 is not a number): a caller keeps the result only when every division has
// cancelled (contains_division).
namespace {
E cancel_round_trip_rec(E e, std::unordered_map<E, E>& memo) {
    if (e->op == CONST || e->op == VAR) return e;
    auto it = memo.find(e);
    if (it != memo.end()) return it->second;
    auto R = [&](E x) { return cancel_round_trip_rec(x, memo); };
    auto product = [&](E a, E b) -> E {
        if (a == b && a->op == FN1 && a->fn == F_SQRT) return a->a;                       // sqrt(A) sqrt(A)
        if (b->op == DIV && b->b == a) return b->a;                                        // a (y / a)
        if (a->op == DIV && a->b == b) return a->a;                                        // (y / b) b
        return mul(a, b);
    };
    E r = nullptr;
    switch (e->op) {
        case ADD: r = add(R(e->a), R(e->b)); break;
        case SUB: r = sub(R(e->a), R(e->b)); break;
        case MUL: r = product(R(e->a), R(e->b)); break;
        case DIV: {
            E n = R(e->a), d = R(e->b);
            r = n == d ? constant(1.0) : div(n, d);
            break;
        }
        case NEG: r = neg(R(e->a)); break;
        case FN1: {
            E a = R(e->a);
            if ((e->fn == F_SIN || e->fn == F_COS) && a->op == FN2 && a->fn == F_ATAN2) {
                E y = a->a, x = a->b;
                E hyp = fn1(F_SQRT, add(product(y, y), product(x, x)));
                r = div(e->fn == F_SIN ? y : x, hyp);
            } else {
                r = fn1(e->fn, a);
            }
            break;
        }
        case FN2: r = fn2(e->fn, R(e->a), R(e->b)); break;
        case SELECT: r = select(R(e->a), R(e->b), R(e->s)); break;
        default: throw std::runtime_error("cancel_round_trip: bad op");
    }
    memo.emplace(e, r);
    return r;
}
bool contains_rec(E e, std::unordered_map<E, bool>& memo, bool (*pred)(E)) {
    if (pred(e)) return true;
    if (e->op == CONST || e->op == VAR) return false;
    auto it = memo.find(e);
    if (it != memo.end()) return it->second;
    bool r = (e->a && contains_rec(e->a, memo, pred)) || (e->b && contains_rec(e->b, memo, pred)) || (e->s && contains_rec(e->s, memo, pred));
    memo.emplace(e, r);
    return r;
}
}  // namespace

E cancel_round_trip(E e) {
    std::unordered_map<E, E> memo;
    return cancel_round_trip_rec(e, memo);
}

bool contains_division_or_angle(E e) {
    std::unordered_map<E, bool> memo;
    return contains_rec(e, memo, [](E x) { return x->op == DIV || (x->op == FN2 && x->fn == F_ATAN2) || (x->op == FN1 && (x->fn == F_SIN || x->fn == F_COS || x->fn == F_TAN ||
                                                                       x->fn == F_ASIN || x->fn == F_ACOS || x->fn == F_ATAN)); });
}

E subst(E e, const std::map<std::string, E>& m) {
    std::unordered_map<E, E> memo;
    return subst_rec(e, m, memo);
}

namespace {
double eval_rec(E e, const std::map<std::string, double>& env, std::unordered_map<E, double>& memo) {
    if (e->op == CONST) return e->c;
    auto it = memo.find(e);
    if (it != memo.end()) return it->second;
    double r = 0;
    switch (e->op) {
        case VAR: {
            auto v = env.find(e->name);
            if (v == env.end()) throw std::runtime_error("eval: unbound variable " + e->name);
            r = v->second;
            break;
        }
        case ADD: r = eval_rec(e->a, env, memo) + eval_rec(e->b, env, memo); break;
        case SUB: r = eval_rec(e->a, env, memo) - eval_rec(e->b, env, memo); break;
        case MUL: r = eval_rec(e->a, env, memo) * eval_rec(e->b, env, memo); break;
        case DIV: r = eval_rec(e->a, env, memo) / eval_rec(e->b, env, memo); break;
        case NEG: r = -eval_rec(e->a, env, memo); break;
        case FN1: r = apply1(e->fn, eval_rec(e->a, env, memo)); break;
        case FN2: r = apply2(e->fn, eval_rec(e->a, env, memo), eval_rec(e->b, env, memo)); break;
        case SELECT:
            r = eval_rec(e->a, env, memo) != 0.0 ? eval_rec(e->b, env, memo) : eval_rec(e->s, env, memo);
            break;
        default: throw std::runtime_error("eval: bad op");
    }
    memo.emplace(e, r);
    return r;
}
}  // namespace

double eval(E e, const std::map<std::string, double>& env) {
    std::unordered_map<E, double> memo;
    return eval_rec(e, env, memo);
}

// ---- device lowering (sym.hpp) ----------------------------------------------------------------------------------------------
namespace {
struct DeviceLowering {
    bool fast_tanh;
    bool changed = false;
    std::unordered_set<E> rooted;            // s of every x / sqrt(s) met in the first pass
    std::unordered_map<E, E> memo;

    // e = position part + literal part (the literal summed through +, - and a literal factor)
    static void split_literal(E e, E& rest, double& lit) {
        if (e->op == CONST) { rest = nullptr; lit = e->c; return; }
        if (e->op == ADD || e->op == SUB) {
            E ra, rb; double la, lb;
            split_literal(e->a, ra, la);
            split_literal(e->b, rb, lb);
            const double sgn = e->op == ADD ? 1.0 : -1.0;
            lit = la + sgn * lb;
            if (!rb) rest = ra;
            else if (!ra) rest = e->op == ADD ? rb : neg(rb);
            else rest = e->op == ADD ? add(ra, rb) : sub(ra, rb);
            return;
        }
        if (e->op == MUL && e->a->op == CONST) {
            E r; double l;
            split_literal(e->b, r, l);
            lit = e->a->c * l;
            rest = r ? mul(e->a, r) : nullptr;
            return;
        }
        if (e->op == NEG) {
            E r; double l;
            split_literal(e->a, r, l);
            lit = -l;
            rest = r ? neg(r) : nullptr;
            return;
        }
        rest = e; lit = 0.0;
    }

    void collect(E e, std::unordered_set<E>& seen) {
        if (!e || !seen.insert(e).second) return;
        if (e->op == DIV && e->b->op == FN1 && e->b->fn == F_SQRT) rooted.insert(e->b->a);
        collect(e->a, seen); collect(e->b, seen); collect(e->s, seen);
    }

    E run(E e) {
        if (e->op == CONST || e->op == VAR) return e;
        auto it = memo.find(e);
        if (it != memo.end()) return it->second;
        E r = nullptr;
        switch (e->op) {
            case ADD: r = add(run(e->a), run(e->b)); break;
            case SUB: r = sub(run(e->a), run(e->b)); break;
            case MUL: {
                E a = run(e->a), b = run(e->b);
                auto trig = [](E x, Fn f) { return x->op == FN1 && x->fn == f; };
                if ((trig(a, F_SIN) || trig(a, F_COS)) && (trig(b, F_SIN) || trig(b, F_COS)) && a->a == b->a && (a->a->deps & ~DEP_CFG) != 0) {
                    r = fn1(a->fn != b->fn ? F_SINCOS : a->fn == F_SIN ? F_SIN2 : F_COS2, a->a);
                    changed = true;
                } else {
                    r = mul(a, b);
                }
                break;
            }
            case NEG: r = neg(run(e->a)); break;
            case DIV:
                if (e->b->op == FN1 && e->b->fn == F_SQRT) {
                    r = mul(run(e->a), fn1(F_RSQRT_FAST, run(e->b->a)));
                    changed = true;
                } else {
                    r = div(run(e->a), run(e->b));
                }
                break;
            case FN1:
                if (e->fn == F_SQRT && rooted.count(e->a)) {
                    E s = run(e->a);
                    r = mul(s, fn1(F_RSQRT_FAST, s));
                    changed = true;
                } else if (e->fn == F_TANH && fast_tanh && (e->deps & ~DEP_CFG) != 0) {
                    const double k = 2.88539008177792681472;   // 2 / ln 2
                    E u = run(e->a), rest = nullptr;
                    double lit = 0.0;
                    split_literal(u, rest, lit);
                    E power;
                    if (rest && lit != 0.0 && std::fabs(k * lit) <= 64.0) power = mul(constant(fl(std::exp2(k * lit))), fn1(F_EXP2_FAST, mul(constant(fl(k)), rest)));
                    else power = fn1(F_EXP2_FAST, mul(constant(fl(k)), u));
                    r = sub(constant(1.0), div(constant(2.0), add(power, constant(1.0))));
                    changed = true;
                } else {
                    r = fn1(e->fn, run(e->a));
                }
                break;
            case FN2: r = fn2(e->fn, run(e->a), run(e->b)); break;
            case SELECT: r = select(run(e->a), run(e->b), run(e->s)); break;
            default: throw std::runtime_error("lower_for_device: bad op");
        }
        memo.emplace(e, r);
        return r;
    }
};
}  // namespace

std::vector<E> lower_for_device(const std::vector<E>& roots, bool fast_tanh, bool* changed) {
    DeviceLowering l;
    l.fast_tanh = fast_tanh;
    std::unordered_set<E> seen;
    for (E r : roots) l.collect(r, seen);
    std::vector<E> out;
    for (E r : roots) out.push_back(l.run(r));
    if (changed) *changed = l.changed;
    return out;
}

std::string const_to_c(double v) {
    float f = (float)v;
    if (std::isinf(f)) return f > 0 ? "INFINITY" : "(-INFINITY)";
    if (std::isnan(f)) return "NAN";
    char buf[64];
    std::snprintf(buf, sizeof(buf), "%.9g", (double)f);
    std::string s = buf;
    if (s.find('.') == std::string::npos && s.find('e') == std::string::npos && s.find("inf") == std::string::npos)
        s += ".0";
    s += "f";
    if (f < 0) s = "(" + s + ")";
    return s;
}

namespace {
thread_local bool g_push_negations = false;
void to_c_rec(E e, const std::unordered_map<E, std::string>* names, std::string& out, bool top);
// -e, the negation carried down to a leaf
void to_c_negated(E e, const std::unordered_map<E, std::string>* names, std::string& out) {
    if (names) {
        auto it = names->find(e);
        if (it != names->end()) { out += "(-" + it->second + ")"; return; }
    }
    switch (e->op) {
        case CONST: out += const_to_c(-e->c); break;
        case NEG: to_c_rec(e->a, names, out, false); break;
        case ADD: out += "("; to_c_negated(e->a, names, out); out += "-"; to_c_rec(e->b, names, out, false); out += ")"; break;
        case SUB: out += "("; to_c_rec(e->b, names, out, false); out += "-"; to_c_rec(e->a, names, out, false); out += ")"; break;
        case MUL: out += "("; to_c_negated(e->a, names, out); out += "*"; to_c_rec(e->b, names, out, false); out += ")"; break;
        case DIV:   // (the device's rendering: quotients as calls - below)
            if (e->a->op == CONST && std::fabs(e->a->c) == 1.0) { out += e->a->c > 0 ? "(-gr_rcp(" : "(gr_rcp("; to_c_rec(e->b, names, out, false); out += "))"; break; }
            out += "gr_div("; to_c_negated(e->a, names, out); out += ","; to_c_rec(e->b, names, out, false); out += ")"; break;
        default: out += "(-"; to_c_rec(e, names, out, false); out += ")"; break;
    }
}
void to_c_rec(E e, const std::unordered_map<E, std::string>* names, std::string& out, bool top) {
    // A graph prints as a tree: what is shared in it and not named is written out at every use, and a script can share exponentially
    // (s = s*s + s in a loop of sixty).  The largest string of the reference's scripts is 0.3 MB; a macro beyond 64 MiB is refused
    // instead of being printed until memory runs out.
    if (out.size() > ((size_t)64 << 20)) throw std::runtime_error("generated expression larger than 64 MiB (a sub-expression shared exponentially often?)");
    if (names && !top) {
        auto it = names->find(e);
        if (it != names->end()) {
            out += it->second;
            return;
        }
    }
    switch (e->op) {
        case CONST: out += const_to_c(e->c); break;
        case VAR: out += e->name; break;
        case DIV:
            // The device's rendering (to_c_negations_pushed: GR_DEVICE_ACCEL*, GR_DEVICE_TEMPORARIES) writes a quotient as gr_div(a, b) and a
            // reciprocal as gr_rcp(b): kernels/metric.hip makes a / b and 1.0f / b of them - the same instructions as the operator - or,
            // in a program built with -DGR_REFINED_RECIPROCALS, the correctly rounded quotient from v_rcp_f32 and a Newton step (round 6).
            // The boundary strings (cl.cl, the oracle) keep the operator.
            if (g_push_negations) {
                if (e->a->op == CONST && std::fabs(e->a->c) == 1.0) { out += e->a->c > 0 ? "gr_rcp(" : "(-gr_rcp("; to_c_rec(e->b, names, out, false); out += e->a->c > 0 ? ")" : "))"; break; }
                out += "gr_div("; to_c_rec(e->a, names, out, false); out += ","; to_c_rec(e->b, names, out, false); out += ")";
                break;
            }
            [[fallthrough]];
        case ADD:
        case SUB:
        case MUL: {
            out += "(";
            to_c_rec(e->a, names, out, false);
            out += e->op == ADD ? "+" : e->op == SUB ? "-" : e->op == MUL ? "*" : "/";
            to_c_rec(e->b, names, out, false);
            out += ")";
            break;
        }
        case NEG:
            if (g_push_negations) { to_c_negated(e->a, names, out); break; }
            out += "(-";
            to_c_rec(e->a, names, out, false);
            out += ")";
            break;
        case FN1:
            out += fn_name(e->fn);
            out += "(";
            to_c_rec(e->a, names, out, false);
            out += ")";
            break;
        case FN2:
            if (e->fn == F_CSQRT_RE || e->fn == F_CSQRT_IM) {
                std::string a, b;
                to_c_rec(e->a, names, a, false);
                to_c_rec(e->b, names, b, false);
                std::string mod = "sqrt(((" + a + "*" + a + ")+(" + b + "*" + b + ")))";
                if (e->fn == F_CSQRT_RE) out += "sqrt(fmax((0.5f*(" + mod + "+" + a + ")),0.0f))";
                else out += "(((" + b + "<0.0f)?(-1.0f):1.0f)*sqrt(fmax((0.5f*(" + mod + "-" + a + ")),0.0f)))";
            } else if (is_cmp(e->fn)) {
                const char* o = e->fn == F_LT ? "<" : e->fn == F_LE ? "<=" : e->fn == F_EQ ? "==" : e->fn == F_GT ? ">" : ">=";
                out += "((";
                to_c_rec(e->a, names, out, false);
                out += o;
                to_c_rec(e->b, names, out, false);
                out += ")?1.0f:0.0f)";
            } else {
                out += fn_name(e->fn);
                out += "(";
                to_c_rec(e->a, names, out, false);
                out += ",";
                to_c_rec(e->b, names, out, false);
                out += ")";
            }
            break;
        case SELECT:
            out += "((";
            to_c_rec(e->a, names, out, false);
            out += "!=0.0f)?";
            to_c_rec(e->b, names, out, false);
            out += ":";
            to_c_rec(e->s, names, out, false);
            out += ")";
            break;
    }
}
}  // namespace

std::string to_c(E e, const std::unordered_map<E, std::string>* names, bool is_definition) {
    std::string out;
    to_c_rec(e, names, out, is_definition);
    return out;
}

// (values, not bits: -(a - b) -> (b - a) and -(a + b) -> ((-a) - b) give +0 where the negated form gives -0 when the operands cancel
// exactly - observable through a division or an atan2 of the result; the device's rendering is used inside the Verlet loop only, where
// the relaxed build carries -fno-signed-zeros anyway.  The flag is put back by a guard: to_c_rec throws on an expression it cannot
// write, and a flag left set would push negations into the strings the next to_c() of this thread writes for cl.cl and the oracle.)
std::string to_c_negations_pushed(E e, const std::unordered_map<E, std::string>* names, bool is_definition) {
    struct guard {
        guard() { g_push_negations = true; }
        ~guard() { g_push_negations = false; }
    } pushed;
    std::string out;
    to_c_rec(e, names, out, is_definition);
    return out;
}

OpCount count_ops(const std::vector<E>& roots) {
    OpCount oc;
    std::unordered_set<E> seen;
    std::vector<E> stack(roots.begin(), roots.end());
    while (!stack.empty()) {
        E e = stack.back();
        stack.pop_back();
        if (!e || !seen.insert(e).second) continue;
        switch (e->op) {
            case CONST:
            case VAR: break;
            case FN1:
                if (e->fn == F_FABS || e->fn == F_SIGN) oc.ops++;
                else if (e->fn == F_SQRT) { oc.ops++; oc.transcendental++; }
                else { oc.ops++; oc.transcendental++; }
                break;
            case FN2:
                oc.ops++;
                if (e->fn == F_ATAN2 || e->fn == F_POW || e->fn == F_FMOD) oc.transcendental++;
                if (e->fn == F_CSQRT_RE || e->fn == F_CSQRT_IM) { oc.ops += 7; oc.transcendental += 2; }
                break;
            default: oc.ops++; break;
        }
        if (e->a) stack.push_back(e->a);
        if (e->b) stack.push_back(e->b);
        if (e->s) stack.push_back(e->s);
    }
    return oc;
}

Temporaries hoist_position_temporaries(const std::vector<E>& roots, const std::string& prefix) {
    // reference counts over the DAG (each parent edge + each root counts once)
    std::unordered_map<E, int> refs;
    std::vector<E> order;  // post-order
    std::unordered_set<E> seen;
    std::function<void(E)> visit = [&](E e) {
        if (!e) return;
        refs[e]++;
        if (!seen.insert(e).second) return;
        visit(e->a);
        visit(e->b);
        visit(e->s);
        order.push_back(e);
    };
    for (E r : roots) visit(r);

    // Parameter-only sub-expressions ($cfg values and constants, no coordinate) that divide or call a function are named even when
    // they are referenced once: the device evaluates the named parameter-only temporaries in double precision (GR_CFG_TEMPORARIES,
    // metric_codegen.cpp), which it cannot do for an expression buried inside a coordinate-dependent one.  Only the outermost such
    // node under a parent that depends on something else (or a root) is taken.
    std::unordered_map<E, bool> delicate_memo;
    std::function<bool(E)> delicate = [&](E e) -> bool {
        if (!e || e->op == CONST || e->op == VAR) return false;
        auto it = delicate_memo.find(e);
        if (it != delicate_memo.end()) return it->second;
        bool d = e->op == DIV || e->op == FN1 || (e->op == FN2 && !is_cmp(e->fn)) || delicate(e->a) || delicate(e->b) || delicate(e->s);
        delicate_memo.emplace(e, d);
        return d;
    };
    std::unordered_set<E> forced;
    auto parameter_only = [](E e) { return e && e->deps == DEP_CFG; };
    for (E e : order) {
        if (parameter_only(e)) continue;
        for (E c : {e->a, e->b, e->s})
            if (parameter_only(c) && delicate(c)) forced.insert(c);
    }
    for (E r : roots)
        if (parameter_only(r) && delicate(r)) forced.insert(r);

    Temporaries t;
    const uint32_t pos_mask = DEP_V1 | DEP_V2 | DEP_V3 | DEP_V4 | DEP_CFG;
    for (E e : order) {
        if (e->op == CONST || e->op == VAR) continue;
        if (e->deps & ~pos_mask) continue;   // depends on velocity or something else
        if (e->deps == 0) continue;          // pure constant expression (should have folded)
        if (refs[e] < 2 && !forced.count(e)) continue;
        if (e->op == NEG && (e->a->op == VAR || e->a->op == CONST)) continue;
        std::string name = prefix + std::to_string(t.defs.size());
        t.defs.emplace_back(name, e);
        t.names.emplace(e, name);
    }
    return t;
}

// ----------------------------------------------------------------------------------------------
// complex

Cx cx(E re) { return Cx{re, constant(0.0)}; }
Cx cx(E re, E im) { return Cx{re, im}; }
Cx cadd(Cx a, Cx b) { return Cx{add(a.re, b.re), add(a.im, b.im)}; }
Cx csub(Cx a, Cx b) { return Cx{sub(a.re, b.re), sub(a.im, b.im)}; }
Cx cneg(Cx a) { return Cx{neg(a.re), neg(a.im)}; }
Cx cconj(Cx a) { return Cx{a.re, neg(a.im)}; }
Cx cmul(Cx a, Cx b) {
    return Cx{sub(mul(a.re, b.re), mul(a.im, b.im)), add(mul(a.re, b.im), mul(a.im, b.re))};
}
E cabs2(Cx a) { return add(mul(a.re, a.re), mul(a.im, a.im)); }
E cabs(Cx a) {
    if (is_zero(a.im)) return fn1(F_FABS, a.re);
    if (is_zero(a.re)) return fn1(F_FABS, a.im);
    return fn1(F_SQRT, cabs2(a));
}
Cx cdiv(Cx a, Cx b) {
    if (is_zero(b.im)) return Cx{div(a.re, b.re), div(a.im, b.re)};
    E d = cabs2(b);
    Cx n = cmul(a, cconj(b));
    return Cx{div(n.re, d), div(n.im, d)};
}
Cx cpowi(Cx a, int n) {
    if (n == 0) return cx(constant(1.0));
    if (n < 0) return cdiv(cx(constant(1.0)), cpowi(a, -n));
    Cx r = a;
    for (int i = 1; i < n; i++) r = cmul(r, a);
    return r;
}
Cx csqrt_real(E a) {
    // sqrt of a real that may be negative: sqrt(|a|) on the real axis for a>=0, imaginary otherwise
    if (a->op == CONST) {
        if (a->c >= 0) return cx(constant(std::sqrt(a->c)));
        return Cx{constant(0.0), constant(std::sqrt(-a->c))};
    }
    E root = fn1(F_SQRT, fn1(F_FABS, a));
    E nonneg = fn2(F_GE, a, constant(0.0));
    return Cx{select(nonneg, root, constant(0.0)), select(nonneg, constant(0.0), root)};
}
Cx csqrt_principal(Cx a) {
    if (is_zero(a.im)) return csqrt_real(a.re);
    // principal branch: sqrt((|z|+re)/2) + i*sign(im)*sqrt((|z|-re)/2), as a differentiable primitive pair
    return Cx{fn2(F_CSQRT_RE, a.re, a.im), fn2(F_CSQRT_IM, a.re, a.im)};
}
Cx csin(Cx a) {
    if (is_zero(a.im)) return cx(fn1(F_SIN, a.re));
    return Cx{mul(fn1(F_SIN, a.re), fn1(F_COSH, a.im)), mul(fn1(F_COS, a.re), fn1(F_SINH, a.im))};
}
Cx ccos(Cx a) {
    if (is_zero(a.im)) return cx(fn1(F_COS, a.re));
    return Cx{mul(fn1(F_COS, a.re), fn1(F_COSH, a.im)), neg(mul(fn1(F_SIN, a.re), fn1(F_SINH, a.im)))};
}

}  // namespace sym
