// Kerr black hole in Kerr-Schild (Cartesian-like, horizon-penetrating) coordinates: g = eta + f l l with the null
// covector l = (1, (r x + a y)/(r^2 + a^2), (r y - a x)/(r^2 + a^2), z / r) and f = rs r^3 / (r^4 + a^2 z^2), where r is
// the positive root of (x^2 + y^2)/(r^2 + a^2) + z^2/r^2 = 1 (Kerr & Schild 1965; Visser, arXiv:0706.0622, eq. 32-34).
// Units: rs = 2M.
function kerr_schild(t, x, y, z)
{
    $cfg.rs.$default = 1;
    $cfg.a.$default = 0.45;

    var rs = $cfg.rs;
    var a = $cfg.a;

    var rho2 = x * x + y * y + z * z - a * a;
    var r2 = 0.5 * (rho2 + CMath.sqrt(rho2 * rho2 + 4 * a * a * z * z));
    var r = CMath.sqrt(r2);

    var f = rs * r2 * r / (r2 * r2 + a * a * z * z);

    var l = [1, (r * x + a * y) / (r2 + a * a), (r * y - a * x) / (r2 + a * a), z / r];
    var eta = [-1, 1, 1, 1];

    var g = [];
    g.length = 16;

    for(var i = 0; i < 4; i++)
    {
        for(var j = 0; j < 4; j++)
        {
            g[i * 4 + j] = f * l[i] * l[j];

            if(i == j)
                g[i * 4 + j] = g[i * 4 + j] + eta[i];
        }
    }

    return g;
}

kerr_schild
