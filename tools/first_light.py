"""First-light GPU check: renders the config metrics, prints stage timings and sanity statistics."""
import sys, time, json, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import geodesic_raytracing_amd as gra
from geodesic_raytracing_amd.pipeline import DeviceBuffer, download, RENDER_DATA_DTYPE, LIGHTRAY_DTYPE

def run(name, w, h, mode, cfg=None, tiled=1, reps=2):
    m = gra.Metric(name)
    feats = m.features(adaptive_sampling=0)
    args = m.argument_string()
    t0 = time.time(); prog = gra.Program(args, 0); t_build = time.time() - t0
    st = gra.RenderState(w, h, 0)
    bg, levels = gra.pack_background(gra.synthetic_background(1024, 512))
    dbg = DeviceBuffer.from_numpy(0, bg)
    out = DeviceBuffer(0, w * h * 16)
    opts = gra.frame_options(mode=mode, tiled=tiled, time_kernels=1, count_attempts=1)
    cam = gra.default_camera()
    vals = m.cfg_values(**(cfg or {}))
    for _ in range(reps):
        st.render(prog, m, cam, out.ptr, (dbg.ptr, 1024, 512, levels), feats, vals, opts)
        st.synchronize()
    ms = st.stage_ms(); att = st.attempts()
    img = out.to_numpy(np.float32, (h, w, 4))
    rd = download(0, st.buffer(gra.BUF_RENDER_DATA), RENDER_DATA_DTYPE, w * h)
    info = {"metric": name, "size": [w, h], "mode": mode, "tiled": tiled, "build_s": round(t_build, 2), "ms": {k: round(v, 3) for k, v in ms.items()},
            "attempts_per_ray": att / (w * h), "terminated1": float((rd["terminated"] == 1).mean()), "terminated2": float((rd["terminated"] == 2).mean()),
            "img_mean": float(img[..., :3].mean()), "img_nan": int(np.isnan(img).sum()),
            "vgpr_trace": prog.kernel_info("gr_do_generic_rays"), "vgpr_fused": prog.kernel_info("gr_trace_fused")}
    print(json.dumps(info), flush=True)
    return img

if __name__ == "__main__":
    os.makedirs("gpurun_out", exist_ok=True)
    run("minkowski", 256, 256, 0)
    run("schwarzschild", 1920, 1080, 0)
    run("schwarzschild", 1920, 1080, 1)
    for mode, tiled in ((0, 0), (0, 1), (1, 1)):
        img = run("kerr_boyer", 3840, 2160, mode, {"a": 0.45}, tiled)
    np.save("gpurun_out/kerr_4k_small.npy", img[::8, ::8].astype(np.float16))
    run("alcubierre", 1920, 1080, 1)
