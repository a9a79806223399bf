// Straight cosmic string along the z axis (Vilenkin 1981): flat space with a wedge of angle 8*pi*mu cut out,
// ds^2 = -dt^2 + drho^2 + (1 - 4 mu)^2 rho^2 dphi^2 + dz^2, in Weyl cylinder coordinates (t, rho, phi, z).
// Locally flat (every Christoffel symbol comes from the cone), but rays passing on either side of the string are
// deflected towards each other by 4*pi*mu each: a double image without magnification.
$cfg.mu.$default = 0.02;

function cosmic_string(t, rho, phi, z)
{
    var cone = 1 - 4 * $cfg.mu;

    return [-1, 1, cone * cone * rho * rho, 1];
}

cosmic_string
