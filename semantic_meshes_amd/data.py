"""`semantic_meshes.data` for the hot path: the Camera POD and mesh holders.

Reference: /root/reference/python/semantic_meshes/src/Data.cu:5-20 registers `Colmap`, `Ply` and
`Camera(rotation, translation, resolution, focal_lengths, principal_point)`.
"""
import math
import os
import struct

import numpy as np

from . import _lib

_FLOATS = (np.dtype(np.float32), np.dtype(np.float64))
_INTS = (np.dtype(np.int32), np.dtype(np.uint32), np.dtype(np.int64), np.dtype(np.uint64))


def _arr(x, what, ndim, kinds, n=None):
    a = np.asarray(x)
    if a.dtype not in kinds:
        # the reference dispatches on exactly these dtypes (python/semantic_meshes/include/Camera.h:21-52)
        if a.dtype.kind in "iu" and kinds is _FLOATS:
            a = a.astype(np.float64)
        elif a.dtype.kind in "iu" and kinds is _INTS:
            a = a.astype(np.int64)
        else:
            raise ValueError("%s: unsupported dtype %s" % (what, a.dtype))
    if a.ndim != ndim or (n is not None and a.shape != n):
        raise ValueError("%s: expected shape %s, got %s" % (what, n, a.shape))
    return a


class Camera:
    """World->camera pinhole camera.

    Mirrors the reference ctor (/root/reference/python/semantic_meshes/include/Camera.h:16-57):
    `rotation[3,3]`, `translation[3]` float/double -> float32 rigid transform Xc = R*X + t;
    `resolution[2]` ints = (W, H); `focal_lengths[2]`, `principal_point[2]` float/double -> float32 ->
    stored as double (PinholeFC<Vector2d, Vector2d>, include/semantic_meshes/render/Camera.h:11).
    """

    def __init__(self, rotation, translation, resolution, focal_lengths, principal_point):
        self.rotation = _arr(rotation, "rotation", 2, _FLOATS, (3, 3)).astype(np.float32)
        self.translation = _arr(translation, "translation", 1, _FLOATS, (3,)).astype(np.float32)
        res = _arr(resolution, "resolution", 1, _INTS, (2,))
        if int(res[0]) <= 0 or int(res[1]) <= 0:
            raise ValueError("resolution must be positive, got %s" % (res,))
        self.resolution = (int(res[0]), int(res[1]))
        self.focal_lengths = _arr(focal_lengths, "focal_lengths", 1, _FLOATS, (2,)).astype(np.float32).astype(np.float64)
        self.principal_point = _arr(principal_point, "principal_point", 1, _FLOATS, (2,)).astype(np.float32).astype(np.float64)
        pod = _lib.CameraPOD()
        pod.rotation[:] = [float(v) for v in self.rotation.reshape(-1)]
        pod.translation[:] = [float(v) for v in self.translation]
        pod.focal[:] = [float(v) for v in self.focal_lengths]
        pod.principal[:] = [float(v) for v in self.principal_point]
        pod.width, pod.height = self.resolution
        self._pod = pod

    @property
    def width(self):
        return self.resolution[0]

    @property
    def height(self):
        return self.resolution[1]

    def __repr__(self):
        return "Camera(resolution=%s, focal_lengths=%s)" % (self.resolution, self.focal_lengths.tolist())


class Mesh:
    """Triangle mesh as the renderer consumes it: float32[V,3] vertices, int32[F,3] faces
    (what data::Ply hands TriangleRenderer, /root/reference/include/semantic_meshes/data/Ply.h:14-15)."""

    def __init__(self, vertices, faces):
        v = np.asarray(vertices)
        f = np.asarray(faces)
        if v.ndim != 2 or v.shape[1] != 3:
            raise ValueError("vertices must be [V,3], got %s" % (v.shape,))
        if f.ndim != 2 or f.shape[1] != 3:
            raise ValueError("faces must be [F,3], got %s" % (f.shape,))
        if f.size and (f.min() < 0 or f.max() >= len(v)):
            raise ValueError("face indices out of range [0, %d)" % len(v))
        self.vertices = np.ascontiguousarray(v, dtype=np.float32)
        self.faces = np.ascontiguousarray(f, dtype=np.int32)


_PLY_TYPES = {
    "char": "i1", "int8": "i1", "uchar": "u1", "uint8": "u1", "short": "i2", "int16": "i2",
    "ushort": "u2", "uint16": "u2", "int": "i4", "int32": "i4", "uint": "u4", "uint32": "u4",
    "float": "f4", "float32": "f4", "double": "f8", "float64": "f8",
}


class Ply(Mesh):
    """`semantic_meshes.data.Ply(path)`: vertex{x,y,z} + face.vertex_indices in, coloured faces out.

    Reference: /root/reference/src/data/Ply.cpp:9-15 (read via tinyply, face key `vertex_indices`) and
    /root/reference/python/semantic_meshes/include/Ply.h:17-51 (`save(path, colors[, binary])`).
    """

    def __init__(self, ply_file):
        v, f = _read_ply(ply_file)
        super().__init__(v, f)
        self.path = ply_file

    def save(self, path, annotation_colors, binary=True):
        colors = np.asarray(annotation_colors)
        if colors.dtype != np.uint8 or colors.ndim != 2 or colors.shape != (len(self.faces), 3):
            # FromClassColors<2>: uint8 rank-2 (Common.h:32-40); failure -> std::invalid_argument (Ply.h:42-45)
            raise ValueError("annotation_colors must be uint8[F,3], got %s %s" % (colors.dtype, colors.shape))
        _write_ply(path, self.vertices, self.faces, colors, binary)


def _read_ply(path):
    with open(path, "rb") as fh:
        if fh.readline().strip() != b"ply":
            raise ValueError("%s is not a PLY file" % path)
        fmt, elements = None, []
        while True:
            line = fh.readline()
            if not line:
                raise ValueError("unterminated PLY header")
            tok = line.decode("ascii", "replace").split()
            if not tok or tok[0] == "comment" or tok[0] == "obj_info":
                continue
            if tok[0] == "format":
                fmt = tok[1]
            elif tok[0] == "element":
                elements.append((tok[1], int(tok[2]), []))
            elif tok[0] == "property":
                if tok[1] == "list":
                    elements[-1][2].append((tok[4], "list", _PLY_TYPES[tok[2]], _PLY_TYPES[tok[3]]))
                else:
                    elements[-1][2].append((tok[2], _PLY_TYPES[tok[1]]))
            elif tok[0] == "end_header":
                break
        verts = faces = None
        if fmt == "ascii":
            rows = fh.read().decode("ascii", "replace").split("\n")
            pos = 0
            for name, count, props in elements:
                chunk = [r.split() for r in rows[pos:pos + count]]
                pos += count
                if name == "vertex":
                    names = [p[0] for p in props]
                    ix = [names.index(k) for k in ("x", "y", "z")]
                    verts = np.array([[float(r[i]) for i in ix] for r in chunk], dtype=np.float32).reshape(-1, 3)
                elif name == "face":
                    faces = _ascii_faces(chunk, props)
        else:
            end = "<" if fmt == "binary_little_endian" else ">"
            for name, count, props in elements:
                if all(len(p) == 2 for p in props):
                    dt = np.dtype([(p[0], end + p[1]) for p in props])
                    data = np.frombuffer(fh.read(dt.itemsize * count), dtype=dt, count=count)
                    if name == "vertex":
                        verts = np.stack([data["x"], data["y"], data["z"]], axis=1).astype(np.float32)
                else:
                    lists = _binary_lists(fh, count, props, end)
                    if name == "face":
                        faces = lists
        if verts is None or faces is None:
            raise ValueError("PLY needs `vertex` x/y/z and `face` vertex_indices elements")
        return verts, faces


def _face_key(props):
    for i, p in enumerate(props):
        if p[0] in ("vertex_indices", "vertex_index") and len(p) == 4:
            return i
    raise ValueError("face element has no vertex_indices list")


def _ascii_faces(chunk, props):
    if len(props) != 1 or _face_key(props) != 0:
        # general case: walk the properties
        out = []
        k = _face_key(props)
        for r in chunk:
            pos = 0
            for i, p in enumerate(props):
                if len(p) == 4:
                    n = int(r[pos])
                    if i == k:
                        if n != 3:
                            raise ValueError("only triangle faces are supported")
                        out.append([int(v) for v in r[pos + 1:pos + 4]])
                    pos += 1 + n
                else:
                    pos += 1
        return np.array(out, dtype=np.int32).reshape(-1, 3)
    a = np.array([[int(v) for v in r[:4]] for r in chunk], dtype=np.int64).reshape(-1, 4)
    if a.size and not np.all(a[:, 0] == 3):
        raise ValueError("only triangle faces are supported")
    return a[:, 1:4].astype(np.int32)


def _binary_lists(fh, count, props, end):
    k = _face_key(props)
    if len(props) == 1:
        _, _, ct, it = props[0]
        # fast path: every face is a triangle -> fixed-size records
        dt = np.dtype([("n", end + ct), ("v", end + it, (3,))])
        pos = fh.tell()
        data = np.frombuffer(fh.read(dt.itemsize * count), dtype=dt, count=count)
        if np.all(data["n"] == 3):
            return data["v"].astype(np.int32)
        fh.seek(pos)
    out = np.empty((count, 3), np.int32)
    for r in range(count):
        for i, p in enumerate(props):
            if len(p) == 4:
                n = int(np.frombuffer(fh.read(np.dtype(p[2]).itemsize), dtype=end + p[2])[0])
                vals = np.frombuffer(fh.read(np.dtype(p[3]).itemsize * n), dtype=end + p[3])
                if i == k:
                    if n != 3:
                        raise ValueError("only triangle faces are supported")
                    out[r] = vals
            else:
                fh.read(np.dtype(p[1]).itemsize)
    return out


def _write_ply(path, vertices, faces, colors, binary):
    header = ["ply", "format %s 1.0" % ("binary_little_endian" if binary else "ascii"),
              "element vertex %d" % len(vertices), "property float x", "property float y", "property float z",
              "element face %d" % len(faces), "property list uchar int vertex_indices",
              "property uchar red", "property uchar green", "property uchar blue", "end_header"]
    with open(path, "wb") as fh:
        fh.write(("\n".join(header) + "\n").encode("ascii"))
        if binary:
            fh.write(np.ascontiguousarray(vertices, dtype="<f4").tobytes())
            rec = np.empty(len(faces), dtype=[("n", "u1"), ("v", "<i4", (3,)), ("c", "u1", (3,))])
            rec["n"], rec["v"], rec["c"] = 3, faces, colors
            fh.write(rec.tobytes())
        else:
            for v in vertices:
                fh.write(("%r %r %r\n" % (float(v[0]), float(v[1]), float(v[2]))).encode("ascii"))
            for f, c in zip(faces, colors):
                fh.write(("3 %d %d %d %d %d %d\n" % (f[0], f[1], f[2], c[0], c[1], c[2])).encode("ascii"))


# ---- COLMAP workspace reader -----------------------------------------------------------
_COLMAP_MODELS = {0: ("SIMPLE_PINHOLE", 3), 1: ("PINHOLE", 4), 2: ("SIMPLE_RADIAL", 4), 3: ("RADIAL", 5), 4: ("OPENCV", 8),
                  5: ("OPENCV_FISHEYE", 8), 6: ("FULL_OPENCV", 12), 7: ("FOV", 5), 8: ("SIMPLE_RADIAL_FISHEYE", 4),
                  9: ("RADIAL_FISHEYE", 5), 10: ("THIN_PRISM_FISHEYE", 12)}
_COLMAP_BY_NAME = {name: (mid, n) for mid, (name, n) in _COLMAP_MODELS.items()}


def _qvec_to_rotation(q):
    w, x, y, z = (float(v) for v in q)
    n = math.sqrt(w * w + x * x + y * y + z * z)
    w, x, y, z = w / n, x / n, y / n, z / n
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)],
                     [2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)],
                     [2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)]], dtype=np.float64)


class Colmap:
    """`semantic_meshes.data.Colmap(workspace)`: cameras.{bin,txt} + images.{bin,txt} -> `getCamera(index | path)`.

    Reference: /root/reference/src/data/Colmap.cpp:7-62 (images sorted by name `:19-21`, lookup by file name
    `:50-59`), /root/reference/include/semantic_meshes/data/Colmap.h:19-25 (Camera{projection, transform,
    resolution}).  Only the two pinhole models the reference's Camera can hold (render/Camera.h:9-12) are
    accepted.  An unknown image name raises KeyError (the reference prints and calls exit(-1), Colmap.cpp:60-61).
    """

    def __init__(self, workspace_path):
        self.path = workspace_path
        self._cameras = self._read_cameras(self._find(workspace_path, "cameras"))
        images = self._read_images(self._find(workspace_path, "images"))
        self._images = sorted(images, key=lambda im: im["name"])
        for im in self._images:
            if im["camera_id"] not in self._cameras:
                raise ValueError("image %r references unknown camera %d" % (im["name"], im["camera_id"]))

    @staticmethod
    def _find(ws, stem):
        for ext in (".bin", ".txt"):
            p = os.path.join(ws, stem + ext)
            if os.path.isfile(p):
                return p
        raise ValueError("no %s.bin / %s.txt in %s" % (stem, stem, ws))

    @staticmethod
    def _camera_entry(model, width, height, params):
        if model == "SIMPLE_PINHOLE":
            f, cx, cy = params[:3]
            return dict(width=int(width), height=int(height), f=(f, f), c=(cx, cy))
        if model == "PINHOLE":
            fx, fy, cx, cy = params[:4]
            return dict(width=int(width), height=int(height), f=(fx, fy), c=(cx, cy))
        # other models may sit in a workspace unused: the error is raised by getCamera() for an image that needs one
        # (the reference's Camera holds only the two pinhole variants, include/semantic_meshes/render/Camera.h:9-12)
        return dict(width=int(width), height=int(height), unsupported=str(model))

    def _read_cameras(self, path):
        cams = {}
        if path.endswith(".txt"):
            for line in open(path):
                tok = line.split()
                if not tok or tok[0].startswith("#"):
                    continue
                cams[int(tok[0])] = self._camera_entry(tok[1], int(tok[2]), int(tok[3]), [float(v) for v in tok[4:]])
        else:
            with open(path, "rb") as fh:
                (n,) = struct.unpack("<Q", fh.read(8))
                for _ in range(n):
                    cid, mid, w, h = struct.unpack("<iiQQ", fh.read(24))
                    if mid not in _COLMAP_MODELS:
                        raise ValueError("unknown COLMAP camera model id %d" % mid)
                    name, np_ = _COLMAP_MODELS[mid]
                    params = struct.unpack("<%dd" % np_, fh.read(8 * np_))
                    cams[cid] = self._camera_entry(name, w, h, params)
        return cams

    @staticmethod
    def _read_images(path):
        images = []
        if path.endswith(".txt"):
            lines = [ln for ln in open(path) if not ln.startswith("#")]
            # every image has two lines: pose + name, then its 2-D points (possibly empty)
            k = 0
            while k < len(lines):
                tok = lines[k].split()
                if len(tok) >= 10:
                    images.append(dict(id=int(tok[0]), q=[float(v) for v in tok[1:5]], t=[float(v) for v in tok[5:8]],
                                       camera_id=int(tok[8]), name=" ".join(tok[9:])))
                    k += 2
                else:
                    k += 1
        else:
            with open(path, "rb") as fh:
                (n,) = struct.unpack("<Q", fh.read(8))
                for _ in range(n):
                    iid = struct.unpack("<I", fh.read(4))[0]
                    q = struct.unpack("<4d", fh.read(32))
                    t = struct.unpack("<3d", fh.read(24))
                    cid = struct.unpack("<I", fh.read(4))[0]
                    name = b""
                    while True:
                        ch = fh.read(1)
                        if ch in (b"\x00", b""):
                            break
                        name += ch
                    (npts,) = struct.unpack("<Q", fh.read(8))
                    fh.seek(24 * npts, 1)
                    images.append(dict(id=iid, q=list(q), t=list(t), camera_id=cid, name=name.decode("utf-8", "replace")))
        return images

    def getImageNum(self):
        return len(self._images)

    def getImageIndex(self, path):
        name = os.path.basename(os.path.normpath(str(path)))
        for i, im in enumerate(self._images):
            if im["name"] == name:
                return i
        raise KeyError("Image with name %s not found in colmap workspace" % name)

    def getCamera(self, image_id):
        """`getCamera(index)` or `getCamera(image_path)` (python/semantic_meshes/include/Colmap.h:15-23)."""
        index = image_id if isinstance(image_id, (int, np.integer)) else self.getImageIndex(image_id)
        if not 0 <= index < len(self._images):
            raise IndexError("image index %d out of range" % index)
        im = self._images[index]
        cam = self._cameras[im["camera_id"]]
        if "unsupported" in cam:
            raise ValueError("COLMAP camera model %s of image %r is not supported (only SIMPLE_PINHOLE and PINHOLE)"
                             % (cam["unsupported"], im["name"]))
        return Camera(_qvec_to_rotation(im["q"]), np.asarray(im["t"], dtype=np.float64),
                      np.asarray([cam["width"], cam["height"]], dtype=np.int64),
                      np.asarray(cam["f"], dtype=np.float64), np.asarray(cam["c"], dtype=np.float64))

    def getCameras(self):
        return [self.getCamera(i) for i in range(len(self._images))]
