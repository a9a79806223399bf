// Alcubierre warp bubble moving along x with speed `velocity`; top-hat shape function of width R, steepness sigma
function alcubierre(t, x, y, z)
{
    $cfg.velocity.$default = 2;
    $cfg.sigma.$default = 1;
    $cfg.R.$default = 2;

    var v = $cfg.velocity;
    var sigma = $cfg.sigma;
    var R = $cfg.R;

    var rs = CMath.fast_length(x - v * t, y, z);
    var shape = (CMath.tanh(sigma * (rs + R)) - CMath.tanh(sigma * (rs - R))) / (2 * CMath.tanh(sigma * R));

    var g = [];
    g.length = 16;

    g[0] = v * v * shape * shape - 1;
    g[1] = -v * shape;
    g[4] = g[1];
    g[5] = 1;
    g[10] = 1;
    g[15] = 1;

    return g;
}

alcubierre
