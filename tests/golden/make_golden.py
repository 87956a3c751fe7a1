#!/usr/bin/env python3
"""Generates tests/golden/*.npz.

The reference has no tests, fixtures or golden vectors, and can be neither compiled nor imported in this
image, so these fixtures are produced by the CPU oracle (oracle/smesh_oracle.cpp), which
restates the in-tree reference lines.  They pin the oracle (and through it the HIP path) against silent
drift; they are DATA (inputs + expected outputs), not reference source.

  cfg1_render.npz   BASELINE cfg1: 10 000-triangle plane, 4 cameras at 640x480: cameras + uint32 index images
                    (run-length encoded) + float32 depth checksums
  cfg1_fuse.npz     fused float32[10000,5] distributions (sum / summax / mul) for seeded probs
  known_answers.npz tiny hand-checkable scenes (KA1-KA12 of SURVEY.md section 4)
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import oracle  # noqa: E402
from semantic_meshes_amd import synth  # noqa: E402


def rle(a):
    flat = a.reshape(-1)
    change = np.flatnonzero(np.diff(flat)) + 1
    starts = np.concatenate([[0], change])
    return flat[starts].astype(np.uint32), np.diff(np.concatenate([starts, [flat.size]])).astype(np.uint32)


def main():
    oracle.set_threads(1)
    oracle.set_accum_double(False)
    mesh, cams, C = synth.scene("cfg1")
    r = oracle.OracleRenderer(mesh.vertices, mesh.faces)
    out = {"vertices": mesh.vertices, "faces": mesh.faces}
    idxs = []
    for k, cam in enumerate(cams):
        idx, depth = r.render(cam)
        idxs.append(idx)
        vals, lens = rle(idx)
        out["cam%d_R" % k] = cam.rotation
        out["cam%d_t" % k] = cam.translation
        out["cam%d_res" % k] = np.asarray(cam.resolution)
        out["cam%d_f" % k] = cam.focal_lengths
        out["cam%d_c" % k] = cam.principal_point
        out["idx%d_vals" % k], out["idx%d_lens" % k] = vals, lens
        d = depth.copy()
        d[~np.isfinite(d)] = 0
        out["depth%d_sum64" % k] = np.float64(d.astype(np.float64).sum())
        out["depth%d_xor" % k] = np.bitwise_xor.reduce(depth.view(np.uint32).reshape(-1))
    np.savez_compressed(os.path.join(HERE, "cfg1_render.npz"), **out)

    fuse = {}
    P = len(mesh.faces)
    for kind in ("sum", "summax", "mul"):
        agg = oracle.OracleAggregator(P, C, kind, 0.5)
        for k, cam in enumerate(cams):
            W, H = cam.resolution
            probs = oracle.synth_probs(W * H, C, synth.probs_seed(7, k), 0.05).reshape(W, H, C)
            agg.add(idxs[k], probs)
        fuse[kind] = agg.get()
    # Mul once more with the float64-accumulating yardstick: the float32 log-domain state of the reference (LogProb<float>,
    # Fusion.cu:85) is itself ~1e-4 away from the exact product after four views; the HIP path's (hi, lo) state is compared with this
    oracle.set_accum_double(True)
    try:
        agg = oracle.OracleAggregator(P, C, "mul", 0.5)
        for k, cam in enumerate(cams):
            W, H = cam.resolution
            probs = oracle.synth_probs(W * H, C, synth.probs_seed(7, k), 0.05).reshape(W, H, C)
            agg.add(idxs[k], probs)
        fuse["mul_float64_state"] = agg.get()
    finally:
        oracle.set_accum_double(False)
    np.savez_compressed(os.path.join(HERE, "cfg1_fuse.npz"), **fuse)
    print("wrote", os.listdir(HERE))


if __name__ == "__main__":
    main()
