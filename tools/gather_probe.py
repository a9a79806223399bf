"""Where does the per-frame time of the multi-GPU path go (1 rank is enough to see it)?
usage: python -m torch.distributed.run --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29512 tools/gather_probe.py"""
import ctypes, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import torch.distributed as dist
import geodesic_raytracing_amd as gra
from geodesic_raytracing_amd import distributed as grd

rank, world = int(os.environ.get("RANK", 0)), int(os.environ.get("WORLD_SIZE", 1))
device = torch.device("cuda", int(os.environ.get("LOCAL_RANK", 0)))
torch.cuda.set_device(device)
if os.environ.get("PROBE_NO_DIST") != "1":
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)
W, H = 3840, 2160
metric = gra.Metric("kerr_boyer", os.path.join(ROOT, "geodesic_raytracing_amd", "scripts"))
cfg = metric.cfg_values(a=0.45)
feats = metric.features(adaptive_sampling=0)
program = gra.pipeline.ProgramManager(metric, 0, feats, cfg).current(wait=True)
state = gra.RenderState(W, H, 0)
bg_np, levels = gra.pack_background(gra.synthetic_background(4096, 2048))
bg = torch.from_numpy(bg_np).to(device)
camera = gra.default_camera()
stream = torch.cuda.current_stream().cuda_stream
plan = grd.StripPlan(H, world, 16)
out = torch.zeros((H, W, 4), dtype=torch.float32, device=device)
gather = grd.FrameGather(plan, W, device, rank, world)
look = ctypes.pointer(camera)


def run(label, body, n=30):
    for _ in range(3):
        body()
    gather.drain(out); torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(n):
        body()
    gather.drain(out); torch.cuda.synchronize()
    print(f"{label:44s} {(time.perf_counter() - t) / n * 1e3:7.3f} ms/frame", flush=True)


def render():
    opts = gra.frame_options(mode=gra.MODE_FUSED, strip_rank=rank, strip_count=world, block_rows=16, compact_out=1)
    opts.next_camera = look
    state.render(program, metric, camera, gather.local_buffer().data_ptr(), (bg.data_ptr(), 4096, 2048, levels), feats, cfg, opts, stream)


run("render only", render)
def render_plain():
    opts = gra.frame_options(mode=gra.MODE_FUSED)
    opts.next_camera = look
    state.render(program, metric, camera, out.data_ptr(), (bg.data_ptr(), 4096, 2048, levels), feats, cfg, opts, stream)


run("render, default options", render_plain)
run("render only (again)", render)
run("render + submit (async gather + assemble)", lambda: (render(), gather.submit(out)))


def gather_only():
    i = gather.slot
    gather.pending[i] = dist.gather(gather.locals[i], gather.parts[i], dst=0, async_op=True)
    gather.slot = (i + 1) % 2
    w = gather.pending[gather.slot]
    if w is not None and w is not True:
        w.wait()
    gather.pending[gather.slot] = None


run("render + async gather, no assemble", lambda: (render(), gather_only()))
run("assemble only", lambda: gather._assemble(gather.parts[0], out))
run("gather only (async, waited next call)", gather_only)
if dist.is_initialized():
    dist.destroy_process_group()
