"""What parking could buy, before it is built: per-ray attempt counts of a frame from the CPU restatement (oracle/restate.cpp; a tool, not
the product), then the lanes a wave issues for under three policies over the same 8x8 tiles:
  * as traced today: 64 x the longest ray of each tile;
  * parking: a tile-wave that has fewer than K rays left after N attempts hands them over; handed-over rays are grouped 64 to a wave in
    the order they were handed over (tiles dearest first) and such a wave parks again under the same rule;
usage: python tools/park_simulation.py <a> <width> <height> [x0 y0 w h]   (window of the frame, default all)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from oracle import build_restate
from oracle.refpipe import OraclePipeline, pack_features
import geodesic_raytracing_amd as gra

a = float(sys.argv[1]); W = int(sys.argv[2]); H = int(sys.argv[3])
win = [int(v) for v in sys.argv[4:8]] if len(sys.argv) >= 8 else [0, 0, W, H]
m = gra.Metric("kerr_boyer", os.path.join(ROOT, "geodesic_raytracing_amd", "scripts"))
cfgv = m.cfg_values(a=a)
pipe = OraclePipeline(build_restate.build(m.argument_string()))
feats = pack_features(adaptive_sampling=0, max_acceleration_change=m.info.max_acceleration_change)
cache = f"/tmp/park/att_{a}_{W}x{H}_{'_'.join(map(str, win))}.npy"
if os.path.exists(cache):
    att = np.load(cache)
else:
    t0 = time.time()
    fr = pipe.frame(W, H, cfgv, feats, use_prepass=False, nthreads=os.cpu_count(), stages="init")
    rays = fr["rays_init"]
    assert len(rays) == W * H
    sel = np.zeros((H, W), dtype=bool)
    sel[win[1]:win[1] + win[3], win[0]:win[0] + win[2]] = True
    idx = (rays["sy"].astype(np.int64) * W + rays["sx"])
    keep = sel.reshape(-1)[idx]
    out = pipe.attempts_per_ray(rays[keep], cfgv, feats, nthreads=os.cpu_count())
    att = np.zeros((H, W), dtype=np.int32)
    att.reshape(-1)[idx[keep]] = out
    att = att[win[1]:win[1] + win[3], win[0]:win[0] + win[2]]
    np.save(cache, att)
    print(f"# traced {keep.sum()} rays in {time.time() - t0:.0f} s", file=sys.stderr)
h, w = att.shape
h8, w8 = h // 8 * 8, w // 8 * 8
tiles = att[:h8, :w8].reshape(h8 // 8, 8, w8 // 8, 8).transpose(0, 2, 1, 3).reshape(-1, 64).astype(np.int64)
useful = int(tiles.sum())
plain = int(64 * tiles.max(axis=1).sum())
print(f"a={a} window {win}: {tiles.shape[0]} tiles, {useful / 1e6:.1f} M attempts, lane utilisation as traced {useful / plain:.4f}")


def simulate(K, N):
    """returns issued lane-attempts"""
    issued = 0
    order = np.argsort(-tiles.max(axis=1), kind="stable")
    queue = []   # (remaining attempts) of parked rays, in parking order
    def run_wave(lengths):
        # lengths: remaining attempts of the wave's rays; returns (wave time, parked remaining list)
        s = np.sort(lengths)[::-1]
        # the wave runs until max(N, the K-th longest ray's end), or until its longest ray ends if that is sooner
        kth = s[K - 1] if len(s) >= K else 0
        t = min(s[0], max(N, kth))
        parked = s[s > t] - t
        return t, parked
    pending = []
    for ti in order:
        t, parked = run_wave(tiles[ti])
        issued += 64 * t
        pending.extend(parked.tolist())
        while len(pending) >= 64:
            batch = np.array(pending[:64]); pending = pending[64:]
            t, parked = run_wave(batch)
            issued += 64 * t
            pending.extend(parked.tolist())
    while pending:
        batch = np.array(pending[:64]); pending = pending[64:]
        if len(pending) == 0:   # the last wave finishes its rays
            issued += 64 * int(batch.max())
            break
        t, parked = run_wave(batch)
        issued += 64 * t
        pending.extend(parked.tolist())
    return issued


for N in (512, 1024, 2048):
    for K in (4, 8, 16, 32):
        iss = simulate(K, N)
        print(f"  park when < {K:2d} rays are left after {N:4d} attempts: utilisation {useful / iss:.4f}, issue time x {iss / plain:.4f}")
