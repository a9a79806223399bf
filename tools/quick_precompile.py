"""Build container: the code objects of the three benched workloads only (Kerr a = 0.45 4K, double Kerr 4K, Alcubierre 8K with redshift;
dynamic + substituted), in parallel - a minute instead of build()'s four, for kernel experiments.  python tools/quick_precompile.py"""
import multiprocessing, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _one(text):
    import geodesic_raytracing_amd as gra
    gra.Program.precompile(text)
    return True


if __name__ == "__main__":
    import geodesic_raytracing_amd as gra
    scripts = os.path.join(ROOT, "geodesic_raytracing_amd", "scripts")
    jobs = []
    for name, cfg, feats in (("kerr_boyer", dict(a=0.45), {}), ("kerr_boyer", dict(a=0.9), {}), ("double_unequal_kerr", {}, {}), ("alcubierre", {}, dict(redshift=1)),
                             ("schwarzschild", {}, {})):
        m = gra.Metric(name, scripts)
        for text in (m.argument_string(), m.argument_string(features=m.features(adaptive_sampling=0, **feats), static=True, cfg_values=m.cfg_values(**cfg))):
            if text not in jobs:
                jobs.append(text)
    with multiprocessing.get_context("spawn").Pool(min(len(jobs), (os.cpu_count() or 2))) as pool:
        pool.map(_one, jobs, chunksize=1)
    print("precompiled", len(jobs))
