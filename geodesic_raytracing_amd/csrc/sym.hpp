// sym.hpp — hash-consed symbolic expression DAG used by the metric code generator.
//
// Role in the reference: the arithmetic behind metric.hpp / equation_context.hpp lives in the
// un-vendored dependency deps/vec (dual_types::value, dual, dual_complex; call sites
// metric.hpp:38-80, 184-244, 247-274 and js_interop.cpp:129-297).  That library is absent from
// the reference checkout, so this is an independent design with the same *outputs*: C expression
// strings over v1..v4 / iv1..iv4 / dv1..dv4 / cfg->NAME / pvN.
//
// Design: every node is interned (structural equality == pointer equality), constructors apply
// local algebraic simplification (constant folding, 0/1 identities, negation hoisting), and
// differentiation is a memoised DAG walk rather than operator-overloaded dual numbers.
#pragma once
#include <cstdint>
#include <map>
#include <string>
#include <unordered_map>
#include <vector>

namespace sym {

enum Op : uint8_t { CONST, VAR, ADD, SUB, MUL, DIV, NEG, FN1, FN2, SELECT };

enum Fn : uint8_t {
    // unary
    F_SIN, F_COS, F_TAN, F_ASIN, F_ACOS, F_ATAN, F_EXP, F_LOG, F_SQRT, F_FABS, F_SINH, F_COSH,
    F_TANH, F_SIGN,
    // device-only forms (lower_for_device; never in a string cl.cl or the oracle compiles): 2^x on v_exp_f32, 1 / sqrt(x) on v_rsq_f32
    F_EXP2_FAST, F_RSQRT_FAST,
    // binary
    F_ATAN2, F_POW, F_FMOD, F_MIN, F_MAX,
    // real / imaginary part of the principal square root of the complex number (a + i b).  Primitive so that its
    // derivative can follow d sqrt(z) = dz / (2 sqrt(z)) (finite for z != 0) instead of differentiating the
    // component formulas, which are 0/0 on the real axis.  Printed inline, no run-time support function needed.
    F_CSQRT_RE, F_CSQRT_IM,
    // comparisons (binary, value is 1.0f / 0.0f)
    F_LT, F_LE, F_EQ, F_GT, F_GE,
    F_NONE,
    // (appended, not inserted: a node's structural hash takes the numeric value of its function - F_NONE for everything that is not a call -
    // and that hash orders the operands of sums and products, i.e. it is part of every generated string and of the fixtures made from them)
    // device-only: sin^2 x, cos^2 x, sin x cos x from one argument reduction - the squares need no quadrant sign, the product one bit of it
    F_SIN2, F_COS2, F_SINCOS
};

struct Node {
    Op op;
    Fn fn;
    double c;            // CONST value
    std::string name;    // VAR name
    const Node* a;
    const Node* b;
    const Node* s;       // third operand (SELECT)
    uint32_t id;         // creation index, deterministic within a process
    uint32_t deps;       // dependency mask, see dep_* below
    uint32_t size;       // tree size estimate (saturating)
    uint64_t shape;      // structural hash (operator, constants, names, operand shapes): orders commutative operands the same
                         // way whatever was built earlier in the process, so a metric's macro string never depends on history
};
typedef const Node* E;

// dependency mask bits
enum : uint32_t {
    DEP_V1 = 1u << 0, DEP_V2 = 1u << 1, DEP_V3 = 1u << 2, DEP_V4 = 1u << 3,
    DEP_IV = 1u << 4,   // any of iv1..iv4
    DEP_DV = 1u << 5,   // any of dv1..dv4
    DEP_CFG = 1u << 6,  // cfg->NAME
    DEP_OTHER = 1u << 7
};

E constant(double v);
E var(const std::string& name);
E add(E a, E b);
E sub(E a, E b);
E mul(E a, E b);
E div(E a, E b);
E neg(E a);
E fn1(Fn f, E a);
E fn2(Fn f, E a, E b);
E select(E cond, E if_true, E if_false);
E powi(E a, int n);

inline bool is_const(E e) { return e->op == CONST; }
inline bool is_zero(E e) { return e->op == CONST && e->c == 0.0; }
inline bool is_one(E e) { return e->op == CONST && e->c == 1.0; }

E diff(E e, const std::string& wrt);
E subst(E e, const std::map<std::string, E>& m);
double eval(E e, const std::map<std::string, double>& env);

// The patterns a coordinate round trip from_polar(to_polar(x)) leaves behind, rewritten away: sin / cos of an atan2 as ratios over
// the hypotenuse, sqrt(A) sqrt(A) = A, x (y / x) = y, x / x = 1.  Not an identity where a hypotenuse vanishes; keep the result
// only if nothing of the kind is left in it (contains_division_or_angle).
E cancel_round_trip(E e);
bool contains_division_or_angle(E e);

// n / (x*x) -> n * ((1/x) * (1/x)): the reciprocal of x is (almost always) needed anyway, and on gfx950 v_rcp_f32
// issues at a quarter of the v_mul_f32 rate.  `memo` carries the rewritten nodes across calls so that several roots
// keep sharing sub-expressions.
E share_reciprocals(E e, std::unordered_map<E, E>& memo);

// The expressions evaluated every Verlet attempt, rewritten for the device's instruction costs (a transcendental issues at a
// quarter of a multiply's rate); the results go into macros of their own (GR_DEVICE_ACCEL*, metric_codegen.cpp) - the strings the
// reference's cl.cl and the CPU oracle compile are not touched:
//   * fast_tanh (the metric's tanh values meet only sums and products, GR_TANH_IN_SUMS_ONLY): tanh u -> 1 - 2 / (2^(k u) + 1),
//     k = 2 / ln 2, and with u = p + c, c a literal with |k c| <= 64: 2^(k p) 2^(k c) - the two tanh of a warp drive's shape
//     function, tanh(sigma (r + R)) and tanh(sigma (r - R)), then share ONE exponential;
//   * x / sqrt(s) -> x rsqrt(s), and where that reciprocal root exists sqrt(s) -> s rsqrt(s): one v_rsq_f32 instead of
//     v_sqrt_f32 + v_rcp_f32 (not an identity at s = 0, where the quotient it replaces is not a number either).
//   * sin x sin x, cos x cos x, sin x cos x -> gr_sin2(x), gr_cos2(x), gr_sincos(x): the same bits (a square has no sign, the product's
//     sign is the quadrant's low bit) without the instructions that put the quadrant's signs on sin x and cos x themselves - all a
//     Boyer-Lindquist chart ever asks of its angle.
// `changed` reports whether any rule fired.
std::vector<E> lower_for_device(const std::vector<E>& roots, bool fast_tanh, bool* changed);

// fully parenthesised C expression (valid OpenCL C, HIP device C++ and host C++); float literals.
// `names` maps node -> identifier for nodes that were hoisted into temporaries.
// `is_definition` prints the body of `e` even when `e` itself has a name (used for "pvN=<body>").
std::string to_c(E e, const std::unordered_map<E, std::string>* names = nullptr, bool is_definition = false);
// the same text with every negation pushed down to a leaf: -(a + b) as ((-a) - b), -(a * b) as ((-a) * b), -(a - b) as (b - a) - the same
// values bit for bit (negation is exact), written so that a compiler folds each into a source modifier of the instruction that consumes it
// instead of negating a finished result with an instruction of its own (the device's copies of the accelerations, GR_DEVICE_ACCEL*)
std::string to_c_negations_pushed(E e, const std::unordered_map<E, std::string>* names = nullptr, bool is_definition = false);
std::string const_to_c(double v);

// operation count of the DAG reachable from `roots` (shared nodes counted once);
// transcendental calls are reported separately.
struct OpCount { int ops = 0; int transcendental = 0; };
OpCount count_ops(const std::vector<E>& roots);

// Hoist position-only (v1..v4, cfg) sub-expressions that are referenced more than once from the
// DAG spanned by `roots` into named temporaries pv0, pv1, ... (topological order).  This is the
// reference's own TEMPORARIES0 mechanism (equation_context.hpp:16-97), applied automatically.
struct Temporaries {
    std::vector<std::pair<std::string, E>> defs;    // in evaluation order
    std::unordered_map<E, std::string> names;
};
Temporaries hoist_position_temporaries(const std::vector<E>& roots, const std::string& prefix = "pv");

// ----------------------------------------------------------------------------------------------
// Complex numbers over E (for the complex-valued scripts: double_kerr*, kerr_newman_*).
struct Cx {
    E re, im;
};
Cx cx(E re);
Cx cx(E re, E im);
Cx cadd(Cx a, Cx b);
Cx csub(Cx a, Cx b);
Cx cmul(Cx a, Cx b);
Cx cdiv(Cx a, Cx b);
Cx cneg(Cx a);
Cx cconj(Cx a);
Cx cpowi(Cx a, int n);
Cx csqrt_principal(Cx a);   // principal square root of a complex number
Cx csqrt_real(E a);         // sqrt of a possibly negative real -> complex
E cabs2(Cx a);              // a * conj(a)
E cabs(Cx a);
Cx csin(Cx a);
Cx ccos(Cx a);

}  // namespace sym
