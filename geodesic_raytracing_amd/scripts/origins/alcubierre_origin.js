// distance to the centre of the warp bubble, which sits at x = velocity * t
function bubble_distance(t, r, theta, phi)
{
    $cfg.velocity.$default = 2;

    var rho = r * CMath.sin(theta);
    var x = rho * CMath.cos(phi) - $cfg.velocity * t;
    var y = rho * CMath.sin(phi);
    var z = r * CMath.cos(theta);

    return CMath.fast_length(x, y, z);
}

bubble_distance
