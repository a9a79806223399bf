// ingoing Eddington-Finkelstein (v, r, theta, phi) -> Schwarzschild time: t = v - r*, r* = r + rs ln|r - rs|
function to_polar(v, r, theta, phi)
{
    var rs = $cfg.rs;
    var tortoise = r + rs * CMath.log(CMath.fabs(r - rs));

    return [v - tortoise, r, theta, phi];
}

to_polar
