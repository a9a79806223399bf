// distance used by the step-size heuristics: the polar radius itself
function distance(t, r, theta, phi)
{
    return r;
}

distance
