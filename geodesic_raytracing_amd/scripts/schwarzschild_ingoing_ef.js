// Schwarzschild black hole in ingoing Eddington-Finkelstein coordinates (v, r, theta, phi): regular at the horizon
//   ds^2 = -(1 - rs/r) dv^2 + 2 dv dr + r^2 dOmega^2
function ingoing(v, r, theta, phi)
{
    $cfg.rs.$default = 1;

    var rs = $cfg.rs;
    var s = CMath.sin(theta);

    var g = [];
    g.length = 16;

    g[0] = -(1 - rs / r);
    g[1] = 1;
    g[4] = 1;
    g[10] = r * r;
    g[15] = r * r * s * s;

    return g;
}

ingoing
