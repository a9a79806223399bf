// Two unequal counter/co-rotating Kerr black holes on the symmetry axis, held apart by a strut, in Weyl
// coordinates (t, rho, phi, z): the extended double-Kerr solution of Manko & Ruiz, Phys. Lett. B 794 (2019) 36,
// https://www.sciencedirect.com/science/article/pii/S0370269319303375 .
// Parameters: masses m1, m2, dimensionless spins fa_i = a_i / m_i, coordinate separation R.
function double_kerr_unequal(t, rho, phi, z)
{
    $cfg.m1.$default = 0.15;
    $cfg.m2.$default = 0.3;
    $cfg.fa1.$default = 1;
    $cfg.fa2.$default = -0.3;
    $cfg.R.$default = 4;

    var I = CMath.i;
    var m1 = $cfg.m1, m2 = $cfg.m2;
    var a1 = $cfg.fa1 * m1;
    var a2 = $cfg.fa2 * m2;
    var R = $cfg.R;

    function sq(x) { return x * x; }
    function cube(x) { return x * x * x; }
    function mod2(w) { return CMath.self_conjugate_multiply(w); }

    var M = m1 + m2;            // total mass
    var J = m1 * a1 + m2 * a2;  // total angular momentum
    var S = R + M;

    // the real root `a` of the cubic that fixes the total-angular-momentum parameter (Cardano form)
    var k = a1 + a2;
    var b = sq(R) - sq(M);
    var c = 2 * S;
    var q = 18 * b * k + 27 * c * J - 9 * c * k * M + 2 * cube(k);
    var p3 = 3 * b + 3 * c * M - sq(k);
    var cbrt2 = CMath.pow(2, 1/3);
    var u = CMath.pow(CMath.sqrt(sq(q) + 4 * cube(p3)) + q, 1/3);
    var a = u / (3 * cbrt2) - cbrt2 * p3 / (3 * u) + k / 3;
    $pin(a);

    var Q = sq(S) + sq(a);
    var d1 = ((m1 * (a1 - a2 + a) + R * a) * Q + m2 * a1 * sq(a)) / sq(Q);
    var d2 = ((m2 * (a2 - a1 + a) + R * a) * Q + m1 * a2 * sq(a)) / sq(Q);

    // half-lengths of the two horizon rods (complex when a constituent is hyper-extreme)
    var s1 = CMath.csqrt(sq(m1) - sq(a1) + 4 * m2 * a1 * d1);
    var s2 = CMath.csqrt(sq(m2) - sq(a2) + 4 * m1 * a2 * d2);
    $pin(s1);
    $pin(s2);

    // distances to the four rod ends
    function rod_distance(offset) { return CMath.psqrt(sq(rho) + sq(z + offset)); }
    var Dp = rod_distance(0.5 * R + s2);
    var Dn = rod_distance(0.5 * R - s2);
    var dp = rod_distance(-0.5 * R + s1);
    var dn = rod_distance(-0.5 * R - s1);

    var mu0 = (S - I * a) / (S + I * a);
    var iMS = I * M * S;

    function lower_weight(sign)
    {
        var num = (sign * s1 - m1 - I * a1) * Q + 2 * a1 * (m1 * a + iMS);
        var den = (sign * s1 - m1 + I * a1) * Q + 2 * a1 * (m1 * a - iMS);
        return num / den / mu0;
    }

    function upper_weight(sign)
    {
        var num = (sign * s2 + m2 - I * a2) * Q - 2 * a2 * (m2 * a - iMS);
        var den = (sign * s2 + m2 + I * a2) * Q - 2 * a2 * (m2 * a + iMS);
        return -mu0 * num / den;
    }

    var rp = lower_weight(1) * dp;
    var rn = lower_weight(-1) * dn;
    var Rp = upper_weight(1) * Dp;
    var Rn = upper_weight(-1) * Dn;
    $pin(rp); $pin(rn); $pin(Rp); $pin(Rn);

    var w1 = s1 * (sq(R) - sq(s1) + sq(s2));
    var w2 = s2 * (sq(R) + sq(s1) - sq(s2));
    var s12 = s1 * s2;

    var A = (sq(R) - sq(s1 + s2)) * (Rp - Rn) * (rp - rn) - 4 * s12 * (Rp - rn) * (Rn - rp);
    var B = 2 * w1 * (Rn - Rp) + 2 * w2 * (rn - rp) + 4 * R * s12 * (Rp + Rn - rp - rn);
    var G = -z * B + w1 * (Rn - Rp) * (rp + rn + R) + w2 * (rn - rp) * (Rp + Rn - R)
            - 2 * s12 * (2 * R * (rp * rn - Rp * Rn - s1 * (rn - rp) + s2 * (Rn - Rp)) + (sq(s1) - sq(s2)) * (rp + rn - Rp - Rn));
    $pin(A); $pin(B); $pin(G);

    var K0 = (Q * (sq(R) - sq(m1 - m2) + sq(a)) - 4 * sq(m1) * sq(m2) * sq(a)) / (m1 * m2 * Q);

    var norm = mod2(A) - mod2(B);
    var AB = CMath.conjugate(A) + CMath.conjugate(B);

    var f = norm / CMath.Real((A + B) * AB);
    var omega = 2 * a - 2 * CMath.Imaginary(G * AB) / norm;
    var e2gamma = norm / CMath.Real(16 * sq(CMath.fabs(s1)) * sq(CMath.fabs(s2)) * sq(K0) * Dp * Dn * dp * dn);
    $pin(f); $pin(omega); $pin(e2gamma);

    // Weyl-Papapetrou line element: -f (dt - omega dphi)^2 + (e^{2 gamma} (drho^2 + dz^2) + rho^2 dphi^2) / f
    var g = [];
    g.length = 16;

    g[0] = -f;
    g[2] = f * omega;
    g[8] = g[2];
    g[5] = e2gamma / f;
    g[10] = sq(rho) / f - f * sq(omega);
    g[15] = e2gamma / f;

    return g;
}

double_kerr_unequal
