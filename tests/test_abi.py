"""CPU tests of the C-ABI boundary: the library loads, exports every symbol include/geodesic_hip.h declares, the
shared struct layouts match the reference's (render_state.hpp:8-29), and host-only entry points behave."""
import ctypes
import os
import re

import numpy as np
import pytest

import geodesic_raytracing_amd as gra
from geodesic_raytracing_amd.pipeline import LIGHTRAY_DTYPE, RENDER_DATA_DTYPE

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols(header="geodesic_hip.h"):
    text = open(os.path.join(ROOT, "include", header)).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(gr_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    """both headers: the contract (geodesic_hip.h - what a maintainer of the reference binds) and the rest of the exports
    (geodesic_hip_internal.h - fused launchers, schedules, measurement hooks); the contract stays small"""
    contract, internal = declared_symbols(), declared_symbols("geodesic_hip_internal.h")
    assert 40 <= len(contract) <= 80 and not set(contract) & set(internal)
    assert len(open(os.path.join(ROOT, "include", "geodesic_hip.h")).read().splitlines()) <= 350
    for header, names in (("geodesic_hip.h", contract), ("geodesic_hip_internal.h", internal)):
        for n in names:
            assert hasattr(gra.lib, n), f"{n} declared in include/{header} but not exported"
    # the nine launchers of the reference's ray kernels and the frame driver are in the contract
    for n in ("gr_cart_to_generic", "gr_init_basis_vectors", "gr_clear_termination_buffer", "gr_init_rays_generic", "gr_do_generic_rays",
              "gr_calculate_singularities", "gr_calculate_render_data", "gr_handle_adaptive_sampling", "gr_render", "gr_render_frame",
              "gr_render_frame_tiled", "gr_metric_load_script", "gr_metric_argument_string", "gr_program_create", "gr_program_manager_current"):
        assert n in contract, n
    for n in ("gr_trace_fused_launch", "gr_order_tiles", "gr_trace_pending", "gr_render_state_attempts", "gr_program_build_key"):
        assert n in internal, n
    # and the Python binding knows a signature for each of them
    assert set(contract) | set(internal) == set(gra.EXPORTED_SYMBOLS)


def test_headers_read_line_by_line():
    """a formatter once folded a declaration, an enum, their comments and a typedef into one run-on paragraph of geodesic_hip.h; it
    compiled, nobody could read it.  Both headers, line-wise: every function declaration, typedef, struct and enum starts its line (after
    at most a closing comment on a line of its own), no line holds two statements of that kind, and no comment opens after code and
    closes before more code follows on the same line."""
    for header in ("geodesic_hip.h", "geodesic_hip_internal.h"):
        inside_comment = False
        for number, line in enumerate(open(os.path.join(ROOT, "include", header)).read().split("\n"), 1):
            where = f"include/{header}:{number}: {line.strip()[:100]}"
            text = line
            if inside_comment:                                   # a block comment that began on an earlier line
                if "*/" not in text:
                    continue
                text = text[text.index("*/") + 2:]
                inside_comment = False
                assert text.strip() == "", "code behind the end of a comment - " + where
            code = re.sub(r"/\*.*?\*/", "", text)                # comments that open and close on the line
            if "/*" in code:
                code, inside_comment = code[:code.index("/*")], True
            # one statement to a line: nothing but white space between a ';' and the end of the code (for-loops there are none)
            assert not re.search(r";\s*\S", code), "two statements on one line - " + where
            # a declaration starts its line (struct members and continuation lines are indented; the rest starts in column 0)
            if re.search(r"\b(?:typedef|enum)\b|\bgr_\w+\s*\(", code) and not line.startswith((" ", "\t", "*")):
                assert re.match(r"(?:typedef|enum|struct|int|long|double|void|unsigned|const|float|size_t|gr_\w+\*?)\b", code), "a declaration that does not start its line - " + where
    # the options struct of the contract keeps to the frame's own description: the look-ahead frames' shares and times and the attempt
    # counter live in gr_frame_tuning (geodesic_hip_internal.h)
    public = open(os.path.join(ROOT, "include", "geodesic_hip.h")).read()
    body = public[public.index("typedef struct gr_frame_options {"):public.index("} gr_frame_options;")]
    for moved in ("next_strip_rank", "next_geodesic_time", "count_attempts"):
        assert moved not in body
    assert len(re.findall(r"^\s{4}[a-z].*?;", body, flags=re.M)) <= 15


def test_struct_layouts():
    assert LIGHTRAY_DTYPE.itemsize == 96          # sizeof(struct lightray), cl.cl:813-824
    assert RENDER_DATA_DTYPE.itemsize == 32       # sizeof(struct render_data), cl.cl:5066-5074
    assert LIGHTRAY_DTYPE.fields["acceleration"][1] == 48 and LIGHTRAY_DTYPE.fields["ku_uobsu"][1] == 64
    assert LIGHTRAY_DTYPE.fields["terminated"][1] == 72 and LIGHTRAY_DTYPE.fields["sy"][1] == 80
    assert RENDER_DATA_DTYPE.fields["z_shift"][1] == 8 and RENDER_DATA_DTYPE.fields["side"][1] == 24
    assert ctypes.sizeof(gra.Features) == 48
    assert ctypes.sizeof(gra.Camera) == 48


def test_feature_defaults_and_packing():
    from oracle.refpipe import pack_features
    f = gra.default_features()
    assert (f.universe_size, f.max_precision_radius, f.field_of_view) == (20.0, 10.0, 90.0)   # main.cpp:1123-1158
    assert f.adaptive_sampling == 1 and f.redshift == 0 and f.min_step == pytest.approx(1e-6)
    assert bytes(f) == pack_features()
    f2 = gra.default_features(redshift=1, universe_size=30.0)
    assert bytes(f2) == pack_features(redshift=1, universe_size=30.0)


def test_default_camera():
    c = gra.default_camera()
    assert list(c.position) == [0.0, 0.0, -4.0, 0.0]                      # main.cpp:669-673
    assert np.allclose(list(c.quat), [-np.sqrt(0.5), 0, 0, np.sqrt(0.5)], atol=1e-7)


def test_errors_are_reported_not_swallowed():
    h = ctypes.c_void_p()
    rc = gra.lib.gr_metric_builtin(b"no_such_metric", ctypes.byref(h))
    assert rc == -1 and b"no_such_metric" in gra.lib.gr_last_error()
    with pytest.raises(gra.GeodesicError):
        gra.Program.precompile("-DTHIS_IS_NOT_A_COMPLETE_MACRO_SET")
    with pytest.raises(gra.GeodesicError):
        gra.Program.precompile("--not-a-define")
    need = ctypes.c_size_t()
    m = gra.Metric("minkowski")
    rc = gra.lib.gr_metric_argument_string(m.handle, None, 0, None, 0, ctypes.create_string_buffer(4), 4, ctypes.byref(need))
    assert rc == -5 and need.value > 1000


def test_device_entry_points_fail_loudly_without_a_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(gra.GeodesicError):
        gra.Program(gra.Metric("minkowski").argument_string(), 0)
    with pytest.raises(gra.GeodesicError):
        gra.RenderState(64, 64, 0)


def test_precompile_produces_gfx950_code_object(tmp_path, monkeypatch):
    monkeypatch.setenv("GR_CACHE_DIR", str(tmp_path))
    gra.Program.precompile(gra.Metric("schwarzschild").argument_string())
    # three code objects per program: the ray kernels (OpenCL's relaxed arithmetic) in two parts - what a fused frame launches, and the
    # reference-shaped sequence + ray compaction, built behind it (round 6: the swap after a parameter change waits for the first only) -
    # and the set-up module (camera, tetrad, the camera's own geodesic: once per frame on one lane, IEEE arithmetic - kernels/camera.hip)
    files = sorted(tmp_path.glob("*.hsaco"))
    setups = [f for f in files if f.name.endswith(".setup.hsaco")]
    parts = [f.read_bytes() for f in files if f not in setups]
    assert len(setups) == 1 and len(parts) == 2
    setup = setups[0].read_bytes()
    frame = [b for b in parts if b"gr_trace_fused" in b]
    rest = [b for b in parts if b"gr_do_generic_rays" in b]
    assert len(frame) == 1 and len(rest) == 1 and frame[0] is not rest[0]
    frame, rest = frame[0], rest[0]
    for blob in (frame, rest, setup):
        assert blob[:4] == b"\x7fELF" and b"gfx950" in blob
    for k in (b"gr_trace_fused", b"gr_render", b"gr_prepass_fused", b"gr_order_tiles", b"gr_trace_fused_lattice", b"gr_adaptive_refine", b"gr_trace_pending", b"gr_trace_pair"):
        assert k in frame, k       # (a dynamic program: adaptive sampling is a run-time feature, its kernels are on the frame's path)
    for k in (b"gr_do_generic_rays", b"gr_do_generic_rays_scheduled", b"gr_init_rays_generic", b"gr_calculate_render_data", b"gr_calculate_singularities",
              b"gr_handle_adaptive_sampling", b"gr_clear_termination_buffer", b"gr_trace_compact", b"gr_sort_tiles_count"):
        assert k in rest and k not in frame, k
    assert b"gr_render" not in rest.replace(b"gr_render_data", b"").replace(b"gr_calculate_render", b"")
    for k in (b"gr_cart_to_generic", b"gr_init_basis_vectors", b"gr_camera_setup", b"gr_get_geodesic_path", b"gr_handle_interpolating_geodesic"):
        assert k in setup and k not in frame and k not in rest
    # a substituted program without adaptive sampling: its adaptive kernels are off the frame's path
    m = gra.Metric("schwarzschild")
    gra.check(gra.lib.gr_program_precompile_frame_path(m.argument_string(features=m.features(adaptive_sampling=0), static=True, cfg_values=m.cfg_values()).encode()))
    newest = max((f for f in tmp_path.glob("*.hsaco") if not f.name.endswith(".setup.hsaco") and f not in files), key=lambda f: f.stat().st_mtime).read_bytes()
    assert b"gr_trace_fused" in newest and b"gr_trace_fused_lattice" not in newest and b"gr_trace_pending" not in newest


def test_both_modules_build_without_the_code_object_manager(tmp_path):
    """where libamd_comgr cannot be loaded (GR_NO_CODE_OBJECT_MANAGER=1 stands in for that) both code objects of a program come from
    hiprtc itself: the ray kernels - used but not cached, they lack the pass over their code - and, since round 5, the set-up module,
    which needs no pass and is cached (until then gr_program_create failed with GR_ERROR_COMPILE where it used to fall back).  In a
    process of its own: the library looks for the code-object manager once."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import geodesic_raytracing_amd as gra\n"
            "gra.Program.precompile(gra.Metric('minkowski').argument_string())\n")
    env = dict(os.environ, GR_CACHE_DIR=str(tmp_path), GR_NO_CODE_OBJECT_MANAGER="1", GR_VERBOSE_BUILD="1", PYTHONPATH=root)
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    assert "set-up module: building through hiprtc" in r.stderr and "could not be applied" in r.stderr
    files = [f.name for f in tmp_path.glob("*.hsaco")]
    assert len(files) == 1 and files[0].endswith(".setup.hsaco")
    blob = (tmp_path / files[0]).read_bytes()
    assert blob[:4] == b"\x7fELF" and b"gr_camera_setup" in blob


def test_background_packing_layout():
    """load_mipped_image (graphics_settings.cpp:152-212): mip i in the top-left corner of slice i, edge replicated"""
    img = gra.synthetic_background(64, 32)
    packed, levels = gra.pack_background(img)
    assert levels == 6 and packed.shape == (6, 32, 64, 4)          # floor(log2(32)) + 1
    assert np.array_equal(packed[0], img)
    m1 = img.reshape(16, 2, 32, 2, 4).astype(np.float32).mean(axis=(1, 3)) / 255.0
    want = (np.clip(m1, 0, 1) * 255).astype(np.uint8)
    assert np.abs(packed[1, :16, :32].astype(int) - want.astype(int)).max() <= 1
    assert np.array_equal(packed[1, :16, 40], packed[1, :16, 31])      # columns right of the mip replicate its last column
    assert np.array_equal(packed[1, 25, :32], packed[1, 15, :32])      # rows below replicate its last row
    assert np.array_equal(packed[5, 20, 50], packed[5, 0, 1])          # 1x2 mip at the last level


def test_strip_block_helpers():
    assert gra.lib.gr_strip_local_blocks(2160, 16, 0, 8) == 17
    assert gra.lib.gr_strip_local_blocks(2160, 16, 7, 8) == 16
    assert sum(gra.lib.gr_strip_local_blocks(2160, 16, r, 8) for r in range(8)) == 135
    assert gra.lib.gr_tiled_slot_count(3840, 2160) == 3840 * 2160
    assert gra.lib.gr_tiled_slot_count(10, 9) == 16 * 16


def test_png_round_trip_and_screenshot_conversion(tmp_path):
    """gr_write_frame_png = the reference's screenshot loop (main.cpp:2791-2800): clamp, linear -> sRGB, clamp, 8 bit"""
    from geodesic_raytracing_amd.render import read_png, write_frame_png
    rng = np.random.RandomState(2)
    frame = rng.uniform(-0.2, 1.3, size=(19, 31, 4)).astype(np.float32)
    path = str(tmp_path / "f.png")
    write_frame_png(path, frame)
    got = read_png(path)
    c = np.clip(frame, 0, 1)
    srgb = np.where(c <= 0.0031308, c * 12.92, 1.055 * np.power(c, 1 / 2.4) - 0.055)
    want = (np.clip(srgb, 0, 1) * 255).astype(np.uint8)
    assert got.shape == (19, 31, 4) and np.abs(got.astype(int) - want.astype(int)).max() <= 1
    # a PNG written by another encoder (all filter types, RGB) reads back exactly
    pil = pytest.importorskip("PIL.Image")
    img = rng.randint(0, 256, size=(23, 17, 3)).astype(np.uint8)
    img[5:15, 3:9] = np.linspace(0, 255, 6).astype(np.uint8)[None, :, None]
    pil.fromarray(img).save(str(tmp_path / "g.png"), optimize=True)
    back = read_png(str(tmp_path / "g.png"))
    assert np.array_equal(back[..., :3], img) and (back[..., 3] == 255).all()
    with pytest.raises(gra.GeodesicError):
        read_png(str(tmp_path / "missing.png"))


def test_cpp_example_builds_against_the_header_alone():
    """examples/render_kerr.cpp uses nothing but include/geodesic_hip.h and the shared library (no HIP headers, no Python):
    it must compile and link; without a GPU it must stop at the first device call with the library's error message"""
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    subprocess.check_call(["make", "-C", os.path.join(root, "examples")], stdout=subprocess.DEVNULL)
    exe = os.path.join(root, "examples", "render_kerr")
    assert os.path.exists(exe)
    r = subprocess.run([exe, os.path.join(root, "geodesic_raytracing_amd", "scripts"), "kerr_boyer", "64", "36", "/tmp/never.png"],
                       capture_output=True, text=True)
    if r.returncode != 0:              # no GPU here: the first device call must say so
        assert "gr_program_manager_create" in r.stderr and "device" in r.stderr.lower()
    # ... and the N-process example (one rank per GPU, RCCL, communicator id through a file)
    exe = os.path.join(root, "examples", "render_tiled")
    assert os.path.exists(exe)
    r = subprocess.run([exe, "--world", "1", "--rank", "0", os.path.join(root, "geodesic_raytracing_amd", "scripts"), "kerr_boyer", "64", "36",
                        "/tmp/never.png"], capture_output=True, text=True)
    if r.returncode != 0:
        assert "gr_program_create" in r.stderr and "device" in r.stderr.lower()
    assert subprocess.run([exe, "--world", "2", "--rank", "1", "a", "b", "8", "8", "c"], capture_output=True).returncode == 2   # no --id-file


def test_pair_kernel_is_built_for_fixed_step_programs_only(tmp_path, monkeypatch):
    """host rule (capi.cpp pair_kernel_applies): gr_trace_pair is compiled into programs that step without the adaptive
    controller and whose loop expressions instantiate on float pairs; adaptive programs get it on request only"""
    import glob
    import os
    import geodesic_raytracing_amd as gra

    def symbols(argument_string, tag):
        d = tmp_path / tag
        d.mkdir()
        monkeypatch.setenv("GR_CACHE_DIR", str(d))
        gra.Program.precompile(argument_string)
        # (the ray kernels are two code objects - the frame path's holds gr_trace_fused)
        (blob,) = [b for b in (open(f, "rb").read() for f in glob.glob(os.path.join(str(d), "*.hsaco")) if not f.endswith(".setup.hsaco")) if b"gr_render" in b.replace(b"gr_render_data", b"")]
        return b"gr_trace_pair" in blob, b"gr_trace_fused" in blob

    fixed = gra.Metric("schwarzschild")
    adaptive = gra.Metric("kerr_boyer")
    assert not fixed.info.adaptive_precision and adaptive.info.adaptive_precision
    assert symbols(fixed.argument_string(), "fixed") == (True, True)
    assert symbols(adaptive.argument_string(), "adaptive") == (False, True)
    monkeypatch.setenv("GR_TRACE_PAIR_BUILD", "0")
    assert symbols(fixed.argument_string(), "never") == (False, True)


def test_assembly_pass_cuts_the_vector_runs_of_the_integrator(tmp_path, monkeypatch):
    """host build path (csrc/codeobject.cpp): the program goes through the code-object manager to assembly text, break_vector_runs
    puts an s_nop after every 8th instruction of a run of vector instructions in the kernels that hold a Verlet loop, and the
    result is assembled in-process.  Checked on the disassembly of the cached code object (no GPU needed): with the pass no run of
    vector instructions in gr_trace_fused is longer than 8, without it (GR_VECTOR_RUN_LIMIT=0) the acceleration is
    one run of > 100; set-up kernels are left as compiled."""
    import glob
    import os
    import re
    import subprocess
    import geodesic_raytracing_amd as gra
    objdump = "/opt/rocm/lib/llvm/bin/llvm-objdump"
    if not os.path.exists(objdump):
        pytest.skip("no llvm-objdump in this image")
    metric = gra.Metric("kerr_boyer")
    args = metric.argument_string(features=metric.features(adaptive_sampling=0), static=True, cfg_values=metric.cfg_values(a=0.45))

    def longest_runs(limit, tag):
        d = tmp_path / tag
        d.mkdir()
        monkeypatch.setenv("GR_CACHE_DIR", str(d))
        monkeypatch.setenv("GR_VECTOR_RUN_LIMIT", str(limit))
        gra.Program.precompile(args)
        paths = [f for f in glob.glob(os.path.join(str(d), "*.hsaco")) if not f.endswith(".setup.hsaco")]
        assert len(paths) == 2   # the kernels of a fused frame, and the others
        text = "".join(subprocess.run([objdump, "-d", "--no-show-raw-insn", path], capture_output=True, text=True, check=True).stdout for path in paths)
        runs, kernel, run = {}, None, 0
        for line in text.splitlines():
            m = re.match(r"^[0-9a-f]+ <(\w+)>:", line)
            if m:
                kernel, run = m.group(1), 0
                continue
            op = line.strip().split(" ")[0] if line.startswith("\t") else ""
            if op.startswith("v_"):
                run += 1
                runs[kernel] = max(runs.get(kernel, 0), run)
            elif op.startswith("s_"):
                run = 0
        return runs

    plain, passed = longest_runs(0, "plain"), longest_runs(8, "pass")
    assert plain["gr_trace_fused"] > 100
    # (the pass counts inside basic blocks, the disassembly has no labels: two runs that meet at a fall-through read as one)
    assert passed["gr_trace_fused"] <= 16 and passed["gr_prepass_fused"] <= 16 and passed["gr_do_generic_rays"] <= 16
    assert passed["gr_render"] == plain["gr_render"]          # not an integrator kernel: as compiled


def _rot(q, v):
    """rot_quat (cl.cl:176-191): v rotated by the unit quaternion q = (x, y, z, w)"""
    qv, w = np.asarray(q[:3], dtype=np.float64), float(q[3])
    t = 2 * np.cross(qv, v)
    return v + w * t + np.cross(qv, t)


def test_origin_on_screen_inverts_the_pixel_direction_map():
    """gr_camera_origin_on_screen (what tile_history shifts the last frame's costs by): a camera placed anywhere along the line a
    pixel looks out through, behind the origin, sees the origin at that pixel - the pixel -> direction map restated from
    cl.cl:2015-2059 (f_stop = (W/2) / tan(fov/2), direction (x - W/2, y - H/2, f_stop) rotated by the camera quaternion)"""
    rng = np.random.default_rng(5)
    W, H = 1920, 1080
    for _ in range(200):
        fov = float(rng.choice([60.0, 90.0, 110.0]))
        q = rng.normal(size=4)
        q /= np.linalg.norm(q)
        x, y = float(rng.uniform(0, W)), float(rng.uniform(0, H))
        f_stop = (W / 2) / np.tan(np.radians(fov) / 2)
        d = _rot(q, np.array([x - W / 2, y - H / 2, f_stop]))
        d /= np.linalg.norm(d)
        cam = gra.default_camera([float(rng.uniform(-1, 1))] + [float(c) for c in -d * rng.uniform(0.5, 30.0)], [float(c) for c in q])
        got = (gra.c_float * 2)()
        assert gra.lib.gr_camera_origin_on_screen(ctypes.byref(cam), fov, W, H, got) == 1
        assert abs(got[0] - x) < 0.05 and abs(got[1] - y) < 0.05
        # looking the other way: nothing to follow
        away = gra.default_camera([0.0] + [float(c) for c in d * 4.0], [float(c) for c in q])
        assert gra.lib.gr_camera_origin_on_screen(ctypes.byref(away), fov, W, H, got) == 0
    on_it = gra.default_camera([0.0, 0.0, 0.0, 0.0])
    assert gra.lib.gr_camera_origin_on_screen(ctypes.byref(on_it), 90.0, W, H, got) == 0


def test_picture_motion_estimate():
    """gr_picture_motion: 0 for the same camera, the focal length times the angle for a turn, times the parallax for a move; a
    camera whose observer speed or flip changed is another picture altogether"""
    W, fov = 3840, 90.0
    f_stop = (W / 2) / np.tan(np.radians(fov) / 2)
    a = gra.default_camera()
    assert gra.lib.gr_picture_motion(ctypes.byref(a), ctypes.byref(a), fov, W) == 0.0
    b = gra.default_camera()
    b.position[1] += 0.02
    assert gra.lib.gr_picture_motion(ctypes.byref(a), ctypes.byref(b), fov, W) == pytest.approx(0.02 / 4.0 * f_stop, rel=1e-3)
    angle = np.radians(1.0)
    turn = np.array([0.0, np.sin(angle / 2), 0.0, np.cos(angle / 2)])
    q = np.array([a.quat[i] for i in range(4)])
    turned = np.concatenate([turn[3] * q[:3] + q[3] * turn[:3] + np.cross(turn[:3], q[:3]), [turn[3] * q[3] - turn[:3] @ q[:3]]])
    c = gra.default_camera(None, [float(v) for v in turned])
    assert gra.lib.gr_picture_motion(ctypes.byref(a), ctypes.byref(c), fov, W) == pytest.approx(angle * f_stop, rel=1e-3)
    minus = gra.default_camera(None, [float(-v) for v in q])   # -q is the same rotation
    assert gra.lib.gr_picture_motion(ctypes.byref(a), ctypes.byref(minus), fov, W) < 1e-2
    d = gra.default_camera()
    d.basis_speed[0] = 0.1
    assert gra.lib.gr_picture_motion(ctypes.byref(a), ctypes.byref(d), fov, W) > 1e8


def test_png_reader_refuses_what_is_not_a_png_it_can_read(tmp_path):
    """files that are missing, empty, something else, cut short, lying about their chunk sizes or their pixels, or in a flavour the reader
    does not do (16-bit, interlaced), and a caller's buffer that is too small: an error code and a message, no crash, no allocation
    of what a header claims (graphics_settings.cpp:152-243 trusts SFML with this; a library cannot)"""
    import struct
    import zlib
    lib = gra.lib

    def read(path, capacity=1 << 20):
        w, h = ctypes.c_int(), ctypes.c_int()
        buf = (ctypes.c_ubyte * capacity)()
        return lib.gr_read_png_rgba8(str(path).encode(), ctypes.byref(w), ctypes.byref(h), buf, capacity), w.value, h.value

    def chunk(tag, data):
        return struct.pack(">I", len(data)) + tag + data + struct.pack(">I", zlib.crc32(tag + data) & 0xFFFFFFFF)

    img = np.random.RandomState(0).randint(0, 255, (16, 24, 4), dtype=np.uint8)
    good = tmp_path / "good.png"
    assert lib.gr_write_png_rgba8(str(good).encode(), img.ctypes.data_as(ctypes.c_void_p), 24, 16) == 0
    assert read(good) == (0, 24, 16)
    assert read(good, 100)[0] == -5                      # GR_ERROR_BUFFER_TOO_SMALL, with the size it needs
    data = good.read_bytes()
    sig = b"\x89PNG\r\n\x1a\n"

    def header(w, h, depth=8, interlace=0):
        return chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, depth, 6, 0, 0, interlace))
    bad = {
        "empty": b"", "gif": b"GIF89a" + b"\0" * 100, "header_cut": data[:20], "pixels_cut": data[:len(data) // 2],
        "a_billion_square": data[:16] + struct.pack(">II", 1 << 30, 1 << 30) + data[24:],
        "no_pixels": data[:16] + struct.pack(">II", 0, 0) + data[24:],
        "not_deflate": sig + header(4, 4) + chunk(b"IDAT", b"notzlibdata") + chunk(b"IEND", b""),
        "fewer_pixels_than_claimed": sig + header(64, 64) + chunk(b"IDAT", zlib.compress(b"\0" * 100)) + chunk(b"IEND", b""),
        "filter_9": sig + header(2, 2) + chunk(b"IDAT", zlib.compress((bytes([9]) + b"\1" * 8) * 2)) + chunk(b"IEND", b""),
        "sixteen_bit": sig + header(2, 2, depth=16) + chunk(b"IDAT", zlib.compress((b"\0" + b"\1" * 16) * 2)) + chunk(b"IEND", b""),
        "interlaced": sig + header(2, 2, interlace=1) + chunk(b"IDAT", zlib.compress((b"\0" + b"\1" * 8) * 2)) + chunk(b"IEND", b""),
        "chunk_of_two_gigabytes": sig + struct.pack(">I", 0x7FFFFFFF) + b"IHDR" + b"\0" * 13,
    }
    for name, content in bad.items():
        path = tmp_path / (name + ".png")
        path.write_bytes(content)
        rc = read(path)[0]
        assert rc < 0 and lib.gr_last_error(), name
    assert read(tmp_path / "not_there.png")[0] < 0
