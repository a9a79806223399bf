// Morris-Thorne wormhole (Ellis throat of radius n) in proper-radial-distance coordinates (t, l, theta, phi);
// l < 0 is the other universe.  See https://arxiv.org/abs/0904.4184 section on wormholes.
function morris_thorne(t, l, theta, phi)
{
    $cfg.n.$default = 1;

    var n = $cfg.n;
    var area = l * l + n * n;

    return [-1, 1, area, area * (CMath.sin(theta) * CMath.sin(theta))];
}

morris_thorne
