// Schwarzschild (t, r, theta, phi) -> ingoing Eddington-Finkelstein advanced time v = t + r*, r* = r + rs ln|r - rs|
function to_ingoing(t, r, theta, phi)
{
    var rs = $cfg.rs;
    var tortoise = r + rs * CMath.log(CMath.fabs(r - rs));

    return [t + tortoise, r, theta, phi];
}

to_ingoing
