// Kerr-Newman black hole (mass, spin, charge) in Boyer-Lindquist coordinates.  Units: rs = 2M, charge length scale rq
// (rq^2 = Q^2 G / (4 pi eps0 c^4)); the horizon exists while a^2 + rq^2 <= (rs/2)^2.
// Newman et al. 1965; line element as in Misner, Thorne & Wheeler, box 33.2, with Delta = r^2 - rs r + a^2 + rq^2.
function kerr_newman(t, r, theta, phi)
{
    $cfg.rs.$default = 1;
    $cfg.a.$default = 0.3;
    $cfg.rq.$default = 0.25;

    var rs = $cfg.rs;
    var a = $cfg.a;
    var q2 = $cfg.rq * $cfg.rq;

    var s2 = CMath.sin(theta) * CMath.sin(theta);
    var c2 = CMath.cos(theta) * CMath.cos(theta);

    var sigma = r * r + a * a * c2;
    var delta = r * r - rs * r + a * a + q2;
    var ra2 = r * r + a * a;

    var g = [];
    g.length = 16;

    // ds^2 = -(Delta/Sigma)(dt - a s2 dphi)^2 + (s2/Sigma)((r^2+a^2) dphi - a dt)^2 + (Sigma/Delta) dr^2 + Sigma dtheta^2
    g[0] = -(delta - a * a * s2) / sigma;
    g[3] = -a * s2 * (ra2 - delta) / sigma;
    g[12] = g[3];
    g[5] = sigma / delta;
    g[10] = sigma;
    g[15] = s2 * (ra2 * ra2 - delta * a * a * s2) / sigma;

    return g;
}

kerr_newman
