// Kerr black hole in Boyer-Lindquist coordinates.  Units: rs = 2M, so |a| = rs/2 is extremal.
function kerr(t, r, theta, phi)
{
    $cfg.rs.$default = 1;
    $cfg.a.$default = -0.5;

    var rs = $cfg.rs;
    var a = $cfg.a;

    var s2 = CMath.sin(theta) * CMath.sin(theta);
    var c2 = CMath.cos(theta) * CMath.cos(theta);

    var sigma = r * r + a * a * c2;
    var delta = r * r - rs * r + a * a;
    var drag = rs * r / sigma;

    var g = [];
    g.length = 16;

    g[0] = -(1 - drag);
    g[5] = sigma / delta;
    g[10] = sigma;
    g[15] = (r * r + a * a + drag * a * a * s2) * s2;
    g[3] = -drag * a * s2;
    g[12] = g[3];

    return g;
}

kerr
