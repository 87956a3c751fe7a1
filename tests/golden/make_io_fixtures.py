#!/usr/bin/env python3
"""Writes the binary I/O fixtures under tests/golden/ byte by byte from the published file-format descriptions -- NOT with the
reader/writer under test (semantic_meshes_amd/data.py is never imported here):

  colmap_bin/cameras.bin, images.bin   COLMAP "Output Format / Binary File Format" (src/base/reconstruction.cc WriteCamerasBinary /
                                       WriteImagesBinary): little endian; cameras: uint64 n, then per camera int32 id, int32 model id
                                       (0 SIMPLE_PINHOLE: f cx cy; 1 PINHOLE: fx fy cx cy; 2 SIMPLE_RADIAL: f cx cy k), uint64 width,
                                       uint64 height, float64 params[]; images: uint64 n, then per image uint32 id, float64 qvec[4]
                                       (w x y z), float64 tvec[3], uint32 camera id, NUL-terminated name, uint64 npoints2D, then
                                       npoints2D x (float64 x, float64 y, int64 point3D id)
  ply/tetra_ascii.ply, tetra_binary_le.ply, tetra_binary_be.ply
                                       PLY 1.0 (Greg Turk, "The PLY Polygon File Format"): header lines, then the elements in
                                       header order; list properties are <count type> <item type>
The text fixtures (colmap_txt/*.txt, the example of COLMAP's documentation plus one PINHOLE image) are committed as typed.
The same scene is stored in colmap_txt and colmap_bin; tests/test_io_fixtures.py holds the expected numbers as literals.
"""
import os
import struct

HERE = os.path.dirname(os.path.abspath(__file__))

cameras = [(1, 0, 3072, 2304, [2559.81, 1536, 1152]),
           (2, 1, 3072, 2304, [2560.56, 2560.56, 1536, 1152]),
           (3, 2, 3072, 2304, [2559.69, 1536, 1152, -0.0218531])]
images = [(1, [0.851773, 0.0165051, 0.503764, -0.142941], [-0.737434, 1.02973, 3.74354], 1, b"P1180141.JPG",
           [(2362.39, 248.498, 58396), (1784.7, 268.254, 59027), (1784.7, 268.254, -1)]),
          (2, [0.851773, 0.0165051, 0.503764, -0.142941], [-0.737434, 1.02973, 3.74354], 1, b"P1180142.JPG",
           [(1190.83, 663.957, 23056), (1258.77, 640.354, 59070)]),
          (7, [0.5, -0.5, 0.5, -0.5], [0.25, -1.5, 6.0], 2, b"A0000007 left.png", [])]

os.makedirs(os.path.join(HERE, "colmap_bin"), exist_ok=True)
with open(os.path.join(HERE, "colmap_bin", "cameras.bin"), "wb") as fh:
    fh.write(struct.pack("<Q", len(cameras)))
    for cid, model, w, h, params in cameras:
        fh.write(struct.pack("<i", cid) + struct.pack("<i", model) + struct.pack("<Q", w) + struct.pack("<Q", h))
        for p in params:
            fh.write(struct.pack("<d", p))
with open(os.path.join(HERE, "colmap_bin", "images.bin"), "wb") as fh:
    fh.write(struct.pack("<Q", len(images)))
    for iid, q, t, cid, name, pts in images:
        fh.write(struct.pack("<I", iid))
        for v in q + t:
            fh.write(struct.pack("<d", v))
        fh.write(struct.pack("<I", cid) + name + b"\x00" + struct.pack("<Q", len(pts)))
        for x, y, pid in pts:
            fh.write(struct.pack("<d", x) + struct.pack("<d", y) + struct.pack("<q", pid))

# ---- PLY: a tetrahedron with extra per-vertex properties a reader must skip, comments, and an extra element ----------------
verts = [(0.0, 0.0, 0.0, 255, 0, 0, 1.5), (1.0, 0.0, 0.0, 0, 255, 0, -2.0), (0.0, 1.0, 0.0, 0, 0, 255, 0.25), (0.0, 0.0, 1.0, 9, 8, 7, 1e3)]
faces = [(0, 2, 1), (0, 1, 3), (1, 2, 3), (2, 0, 3)]
edges = [(0, 1), (2, 3)]


def header(fmt):
    return ("ply\nformat %s 1.0\ncomment made by hand from the PLY specification\nobj_info tetrahedron fixture\n"
            "element vertex 4\nproperty float x\nproperty float y\nproperty float z\nproperty uchar red\nproperty uchar green\n"
            "property uchar blue\nproperty double quality\nelement face 4\nproperty list uchar int vertex_indices\n"
            "element edge 2\nproperty int vertex1\nproperty int vertex2\nend_header\n" % fmt).encode("ascii")


os.makedirs(os.path.join(HERE, "ply"), exist_ok=True)
with open(os.path.join(HERE, "ply", "tetra_ascii.ply"), "wb") as fh:
    fh.write(header("ascii"))
    for v in verts:
        fh.write(("%g %g %g %d %d %d %r\n" % v).encode("ascii"))
    for f in faces:
        fh.write(("3 %d %d %d\n" % f).encode("ascii"))
    for e in edges:
        fh.write(("%d %d\n" % e).encode("ascii"))
for name, end in (("tetra_binary_le.ply", "<"), ("tetra_binary_be.ply", ">")):
    with open(os.path.join(HERE, "ply", name), "wb") as fh:
        fh.write(header("binary_little_endian" if end == "<" else "binary_big_endian"))
        for x, y, z, r, g, b, q in verts:
            fh.write(struct.pack(end + "fff", x, y, z) + struct.pack("BBB", r, g, b) + struct.pack(end + "d", q))
        for f in faces:
            fh.write(struct.pack("B", 3) + struct.pack(end + "iii", *f))
        for e in edges:
            fh.write(struct.pack(end + "ii", *e))
print("wrote colmap_bin/{cameras,images}.bin and ply/tetra_{ascii,binary_le,binary_be}.ply")
