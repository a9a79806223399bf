// (t, x, y, z) -> (t, r, theta, phi); theta from the z axis, phi in the x-y plane
function to_polar(t, x, y, z)
{
    var rho2 = x * x + y * y;

    return [t, CMath.sqrt(rho2 + z * z), CMath.atan2(CMath.sqrt(rho2), z), CMath.atan2(y, x)];
}

to_polar
