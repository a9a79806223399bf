// Schwarzschild black hole in (t, r, theta, phi) with the Schwarzschild radius as a run-time parameter; integrated with the
// adaptive step controller (BASELINE.json configs[1]: "schwarzschild, adaptive Verlet")
function schwarzschild_adaptive(t, r, theta, phi)
{
    $cfg.rs.$default = 1;

    var lapse = 1 - $cfg.rs / r;
    var s = CMath.sin(theta);

    return [-lapse, 1 / lapse, r * r, r * r * s * s];
}

schwarzschild_adaptive
