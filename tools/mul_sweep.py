"""One-off sweep of the Mul aggregator over every triangle-order kernel against the float64-accumulating oracle: random triangle
soups (k_fuse_tri, _any, _wide and their big-triangle waves, by class count) and texel renderers (k_fuse_texel, _multi), views fused
one by one or in groups, optional weights.  Reports the relative tolerance each scene would have needed on get().
usage (GPU box): python tools/mul_sweep.py [first_seed] [count]"""
import os, sys, time, types
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import semantic_meshes_amd as sm
from semantic_meshes_amd.device import to_device
from oracle import oracle
from helpers import random_probs
from test_gpu_fuzz import _soup, _camera

oracle.set_threads(1)
first, count = (int(sys.argv[1]) if len(sys.argv) > 1 else 0), (int(sys.argv[2]) if len(sys.argv) > 2 else 200)
worst, fails, t0 = {}, 0, time.time()
for seed in range(first, first + count):
    rng = np.random.default_rng(31000 + seed)
    texel = rng.random() < 0.3
    nfaces = int(rng.choice([60, 500, 3000]))
    spread = float(rng.choice([0.004, 0.02, 0.08, 0.4]))
    verts, faces = _soup(rng, 0, nfaces, spread)
    W, H = int(rng.choice([64, 160, 333])), int(rng.choice([48, 120, 257]))
    C = int(rng.choice([2, 7, 19, 40] if texel else [3, 19, 33, 40, 47, 48, 49, 64, 100, 127, 128, 150, 257, 300]))
    iew = float(rng.choice([0.0, 0.5, 1.0]))
    nviews = int(rng.choice([1, 3, 5, 9]))
    cams = [_camera(sm, rng, W, H) for _ in range(nviews)]
    mesh = types.SimpleNamespace(vertices=verts, faces=faces)
    if texel:
        tpp = float(rng.choice([0.1, 0.5, 1.5]))
        r, o = sm.render.texels(mesh, cams, tpp), oracle.OracleRenderer(verts, faces, cams, tpp)
    else:
        r, o = sm.render.triangles(mesh), oracle.OracleRenderer(verts, faces)
    P = r.getPrimitivesNum()
    if P == 0:
        continue
    probs, weights = [], []
    for cam in cams:
        p = random_probs(rng, W, H, C, zero_fraction=0.1)
        probs.append(np.where(p.sum(-1, keepdims=True) > 0, np.maximum(p, 1e-3), 0).astype(np.float32))
        weights.append(rng.random((W, H), dtype=np.float32) if rng.random() < 0.4 else None)
    agg = sm.fusion.MeshAggregator(P, C, "mul", iew)
    grouped = rng.random() < 0.5
    if grouped:
        use_w = any(w is not None for w in weights)
        agg.fuse_views(r, cams, [to_device(p) for p in probs], [None if w is None else to_device(w) for w in weights] if use_w else None)
    else:
        for cam, p, w in zip(cams, probs, weights):
            agg.fuse_view(r, cam, p, w)
    kernel = sm._lib.last_fuse_kernel()
    oracle.set_accum_double(True)
    oagg = oracle.OracleAggregator(P, C, "mul", iew)
    for cam, p, w in zip(cams, probs, weights):
        oagg.add(o.render(cam)[0], p, w)
    got, want = agg.get().astype(np.float64), oagg.get().astype(np.float64)
    oracle.set_accum_double(False)
    err = np.abs(got - want)
    need = float(((err - 1e-6) / np.maximum(np.abs(want), 1e-300)).max())
    worst[kernel] = max(worst.get(kernel, 0.0), need)
    if need > 2e-5:
        fails += 1
        print("seed %d: %s texel=%d %dx%d P=%d C=%d iew=%.1f views=%d grouped=%d needs rtol %.2e" % (seed, kernel, texel, W, H, P, C, iew, nviews, grouped, need), flush=True)
print("%d scenes, %d beyond 2e-5; tolerance needed per kernel: %s; %.0f s" % (count, fails, {k: "%.1e" % v for k, v in worst.items()}, time.time() - t0), flush=True)
