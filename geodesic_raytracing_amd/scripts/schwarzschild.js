// Schwarzschild black hole in (t, r, theta, phi), Schwarzschild radius rs = 1, c = 1
function schwarzschild(t, r, theta, phi)
{
    var lapse = 1 - 1 / r;
    var r2 = r * r;

    return [-lapse, 1 / lapse, r2, r2 * CMath.sin(theta) * CMath.sin(theta)];
}

schwarzschild
