// (t, r, theta, phi) is already the renderer's polar chart
function identity(t, r, theta, phi)
{
    return [t, r, theta, phi];
}

identity
