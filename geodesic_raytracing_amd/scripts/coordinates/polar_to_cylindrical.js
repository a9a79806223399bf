// (t, r, theta, phi) -> Weyl cylinder (t, rho, phi, z)
function to_cylinder(t, r, theta, phi)
{
    return [t, r * CMath.sin(theta), phi, r * CMath.cos(theta)];
}

to_cylinder
