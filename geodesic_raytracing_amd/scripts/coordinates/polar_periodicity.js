// coordinate periods of (t, r, theta, phi); 0 = not periodic
function periods(t, r, theta, phi)
{
    return [0, 0, Math.PI, 2 * Math.PI];
}

periods
