// Flat space with a plane wave in its lapse: g_tt = -(1 + amplitude sin(frequency t + x)), Cartesian chart.  A small time-dependent
// metric whose expressions take the sine and cosine of a COORDINATE THAT IS NOT AN ANGLE: a camera at t = 9000 puts the argument beyond
// the range of the Verlet loop's polynomial sin / cos (|x| < 8192, kernels/metric.hip), so every ray goes through the loop's libm rescue
// (tests/golden: time_ripple_late, time_ripple_late_reparameterised).
function time_ripple(t, x, y, z)
{
    $cfg.amplitude.$default = 0.05;
    $cfg.frequency.$default = 1;

    var lapse = 1 + $cfg.amplitude * CMath.sin($cfg.frequency * t + x);

    return [-lapse, 1, 1, 1];
}

time_ripple
