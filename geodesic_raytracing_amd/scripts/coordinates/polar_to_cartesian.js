// (t, r, theta, phi) -> (t, x, y, z)
function to_cartesian(t, r, theta, phi)
{
    var rho = r * CMath.sin(theta);

    return [t, rho * CMath.cos(phi), rho * CMath.sin(phi), r * CMath.cos(theta)];
}

to_cartesian
