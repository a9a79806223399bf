// Weyl cylinder (t, rho, phi, z) -> (t, r, theta, phi)
function to_polar(t, rho, phi, z)
{
    return [t, CMath.sqrt(rho * rho + z * z), CMath.atan2(rho, z), phi];
}

to_polar
