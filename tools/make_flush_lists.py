"""Builds accumulator-row access lists from an oracle-rendered cfg2 view for tools/flush_replay.hip.
Development experiment: prices candidate flush orders of the scatter-add before writing the kernel."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import oracle  # noqa: E402
from semantic_meshes_amd import synth  # noqa: E402

out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "flush_lists")
os.makedirs(out, exist_ok=True)
mesh, cams, C = synth.scene("cfg2")
r = oracle.OracleRenderer(mesh.vertices, mesh.faces)
BG = 0xFFFFFFFF


def runs_1d(idx_flat, tile):
    res = []
    for b in range(0, len(idx_flat), tile):
        seg = idx_flat[b:b + tile]
        heads = np.ones(len(seg), bool)
        heads[1:] = seg[1:] != seg[:-1]
        res.append(seg[heads])
    a = np.concatenate(res)
    return a[a != BG]


def tiles_2d(idx, tx, ty):
    W, H = idx.shape
    lists = []
    for x0 in range(0, W, tx):
        blk = idx[x0:x0 + tx]
        for y0 in range(0, H, ty):
            u = np.unique(blk[:, y0:y0 + ty])
            lists.append(u[u != BG])
    return lists


for view in (0, 66):
    idx, _ = r.render(cams[view])
    flat = idx.reshape(-1)
    T = np.unique(flat)
    T = T[T != BG]
    stats = {"T": len(T)}
    l1 = runs_1d(flat, 256)
    l1.astype(np.uint32).tofile(os.path.join(out, "v%d_runs1d.u32" % view))
    stats["runs1d"] = len(l1)
    T.astype(np.uint32).tofile(os.path.join(out, "v%d_ideal.u32" % view))
    # 4 x 16 strips in strip order, groups in first-pixel order (what the wave-per-strip kernel emits)
    W, H = idx.shape
    res = []
    for x0 in range(0, W, 4):
        blk = idx[x0:x0 + 4]
        for y0 in range(0, H, 16):
            sub = blk[:, y0:y0 + 16].reshape(-1)
            _, first = np.unique(sub, return_index=True)
            u = sub[np.sort(first)]
            res.append(u[u != BG])
    strips = np.concatenate(res)
    strips.astype(np.uint32).tofile(os.path.join(out, "v%d_strips4x16.u32" % view))
    stats["strips4x16"] = len(strips)
    for (tx, ty) in ((16, 16),):
        lists = tiles_2d(idx, tx, ty)
        allr = np.concatenate(lists)
        cnt = np.bincount(allr, minlength=len(mesh.faces))
        excl = np.concatenate([l[cnt[l] == 1] for l in lists])
        strad = np.concatenate([l[cnt[l] > 1] for l in lists])
        allr.astype(np.uint32).tofile(os.path.join(out, "v%d_t%dx%d_all.u32" % (view, tx, ty)))
        # hybrid list: high bit marks rows that must use atomics (primitive seen by more than one tile)
        hyb = np.concatenate([np.where(cnt[l] > 1, l | 0x80000000, l) for l in lists]).astype(np.uint32)
        hyb.tofile(os.path.join(out, "v%d_t%dx%d_hyb.u32" % (view, tx, ty)))
        stats["t%dx%d" % (tx, ty)] = (len(allr), len(excl), len(strad))
    print(view, stats)
