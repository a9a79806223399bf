// flat spacetime in Cartesian coordinates, signature (-,+,+,+)
function flat(t, x, y, z)
{
    return [-1, 1, 1, 1];
}

flat
