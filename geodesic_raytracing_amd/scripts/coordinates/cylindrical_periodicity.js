// coordinate periods of (t, rho, phi, z); 0 = not periodic
function periods(t, rho, phi, z)
{
    return [0, 0, 2 * Math.PI, 0];
}

periods
