"""Synthetic benchmark scenes: grid height-field meshes, a ring of look-at pinhole
cameras, and hashed class-probability images generated directly in HBM.

Everything is a closed-form function of integer indices (no RNG state), so every rank, the CPU oracle
and the GPU see identical inputs.
"""
import ctypes
import math

import numpy as np

from . import _lib
from .data import Camera, Mesh
from .device import DeviceArray, DeviceBuffer

# BASELINE.json configs: (quads_a, quads_b, width, height, classes, views)
CONFIGS = {
    "cfg1": dict(a=100, b=50, width=640, height=480, classes=5, views=4),
    "cfg2": dict(a=1000, b=500, width=1920, height=1080, classes=19, views=200),
    "cfg4": dict(a=2500, b=1000, width=1296, height=968, classes=40, views=1000, texels=True),   # render.texels path
    # cfg4's mesh and cameras with texels_per_pixel 5: the sub-pixel triangles get r = 3 (six texels each, 30 M primitives, a 4.8 GB
    # accumulator) -- the texel shader (TexturedTriangleRenderer.h:31-41) is exercised with r > 1, which cfg4's default 0.1 never does
    "cfg4t": dict(a=2500, b=1000, width=1296, height=968, classes=40, views=1000, texels=True, texels_per_pixel=5.0),
    "cfg5": dict(a=5000, b=2000, width=4096, height=2160, classes=150, views=500),
}


def grid_mesh(a, b, extent=10.0, relief=0.15):
    """Regular grid height-field with exactly 2*a*b triangles and (a+1)*(b+1) vertices.

    Vertex (i,j) = (i*s, j*s, A*sin(k1*i)*cos(k2*j)) centred on the origin; faces (v00,v10,v01), (v10,v11,v01).
    """
    s = extent / a
    i = np.arange(a + 1, dtype=np.float64)[:, None]
    j = np.arange(b + 1, dtype=np.float64)[None, :]
    x = np.broadcast_to(i * s - 0.5 * a * s, (a + 1, b + 1))
    y = np.broadcast_to(j * s - 0.5 * b * s, (a + 1, b + 1))
    z = relief * extent * 0.1 * np.sin(i * (2 * math.pi * 3.0 / a)) * np.cos(j * (2 * math.pi * 2.0 / b))
    vertices = np.stack([x, y, z], axis=-1).reshape(-1, 3).astype(np.float32)
    ii, jj = np.meshgrid(np.arange(a), np.arange(b), indexing="ij")
    v00 = (ii * (b + 1) + jj).reshape(-1)
    v10 = v00 + (b + 1)
    v01 = v00 + 1
    v11 = v10 + 1
    faces = np.empty((a * b, 2, 3), np.int32)
    faces[:, 0, 0], faces[:, 0, 1], faces[:, 0, 2] = v00, v10, v01
    faces[:, 1, 0], faces[:, 1, 1], faces[:, 1, 2] = v10, v11, v01
    return Mesh(vertices, faces.reshape(-1, 3))


def look_at(eye, target, up=(0.0, 0.0, 1.0)):
    """World->camera rigid transform (R, t) with +x right, +y down, +z forward (depth)."""
    eye, target, up = (np.asarray(v, dtype=np.float64) for v in (eye, target, up))
    fwd = target - eye
    fwd /= np.linalg.norm(fwd)
    right = np.cross(fwd, up)
    right /= np.linalg.norm(right)
    down = np.cross(fwd, right)
    R = np.stack([right, down, fwd], axis=0)
    t = -R @ eye
    return R.astype(np.float32), t.astype(np.float32)


def ring_camera(k, views, width, height, extent=10.0, radius_scale=0.62):
    """View k of `views`: on a ring around the mesh centre, elevation sweeping 35..60 degrees,
    pinhole f = 0.8*W, c = (W/2, H/2)."""
    theta = 2.0 * math.pi * k / max(views, 1)
    elev = math.radians(35.0 + 25.0 * (0.5 + 0.5 * math.sin(2.0 * math.pi * (k * 7 % max(views, 1)) / max(views, 1))))
    rho = radius_scale * extent
    eye = (rho * math.cos(theta) * math.cos(elev), rho * math.sin(theta) * math.cos(elev), rho * math.sin(elev))
    R, t = look_at(eye, (0.0, 0.0, 0.0))
    f = 0.8 * width
    return Camera(R, t, np.asarray([width, height]), np.asarray([f, f], dtype=np.float64),
                  np.asarray([width / 2.0, height / 2.0], dtype=np.float64))


def scene(name):
    """(mesh, [cameras], classes) of a BASELINE.json config."""
    cfg = CONFIGS[name]
    mesh = grid_mesh(cfg["a"], cfg["b"])
    cams = [ring_camera(k, cfg["views"], cfg["width"], cfg["height"]) for k in range(cfg["views"])]
    return mesh, cams, cfg["classes"]


def probs_seed(base_seed, view):
    return (int(base_seed) * 0x9E3779B1 + int(view) * 0x85EBCA6B + 0x5EED) & 0xFFFFFFFFFFFFFFFF


def device_probs(width, height, classes, seed, zero_fraction=0.0, device=0, out=None):
    """float32 (W,H,C) class probabilities generated in HBM (kernel k_synth_probs); rows sum to 1,
    `zero_fraction` of the pixels are all-zero don't-care rows."""
    n = width * height
    if out is None:
        buf = DeviceBuffer(n * classes * 4, device)
        out = buf.view((width, height, classes), np.float32)
    _lib.check(_lib.lib().smesh_synth_probs(ctypes.c_void_p(out.ptr), n, classes, seed & 0xFFFFFFFFFFFFFFFF,
                                           float(zero_fraction), device, _lib.MEM_DEVICE))
    return out
