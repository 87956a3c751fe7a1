"""The COLMAP and PLY readers against fixtures authored byte by byte from the published format descriptions
(tests/golden/make_io_fixtures.py and the typed text files; nothing here was written by the code under test), with the
expected numbers as literals.  Reference behaviour: /root/reference/src/data/Colmap.cpp:7-23 (images sorted by name),
:50-59 (lookup by file name), /root/reference/include/semantic_meshes/data/Colmap.h:19-25; /root/reference/src/data/Ply.cpp:9-15
(vertex x,y,z + face vertex_indices), /root/reference/python/semantic_meshes/include/Ply.h:17-36 (coloured write)."""
import os

import numpy as np
import pytest

import semantic_meshes_amd as sm

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

# rotation of qvec (w, x, y, z) = (0.851773, 0.0165051, 0.503764, -0.142941), from scipy.spatial.transform.Rotation
R_P118 = np.array([[0.4515793832, 0.2601359298, 0.8534666711],
                   [-0.2268772277, 0.9585909027, -0.172134264],
                   [-0.8629036935, -0.1158998675, 0.4918988071]])
R_A7 = np.array([[0.0, 0.0, 1.0], [-1.0, 0.0, 0.0], [0.0, -1.0, 0.0]])      # qvec (0.5, -0.5, 0.5, -0.5)


@pytest.mark.parametrize("kind", ["colmap_txt", "colmap_bin"])
def test_colmap_workspace_fixture(kind):
    ws = sm.data.Colmap(os.path.join(GOLDEN, kind))
    assert ws.getImageNum() == 3
    # sorted by NAME, not by id (Colmap.cpp:19-21): "A0000007 left.png" < "P1180141.JPG" < "P1180142.JPG"
    assert ws.getImageIndex("/some/dir/A0000007 left.png") == 0
    assert ws.getImageIndex("P1180141.JPG") == 1 and ws.getImageIndex("images/P1180142.JPG") == 2
    cam = ws.getCamera("/data/scene/images/P1180141.JPG")          # lookup by file name (Colmap.cpp:50-59)
    assert cam.resolution == (3072, 2304)
    np.testing.assert_allclose(cam.rotation, R_P118, atol=2e-7)   # float32 (python/.../Camera.h:16-57)
    np.testing.assert_allclose(cam.translation, [-0.737434, 1.02973, 3.74354], rtol=1e-7)
    assert cam.rotation.dtype == np.float32 and cam.translation.dtype == np.float32
    np.testing.assert_array_equal(cam.focal_lengths, np.float32([2559.81, 2559.81]).astype(np.float64))   # SIMPLE_PINHOLE: f, f
    np.testing.assert_array_equal(cam.principal_point, [1536.0, 1152.0])
    same = ws.getCamera(2)                                          # the second image has the same pose
    np.testing.assert_array_equal(same.rotation, cam.rotation)
    pin = ws.getCamera(0)                                           # PINHOLE camera 2, name with a blank in it
    np.testing.assert_allclose(pin.rotation, R_A7, atol=1e-7)
    np.testing.assert_array_equal(pin.translation, np.float32([0.25, -1.5, 6.0]))
    np.testing.assert_array_equal(pin.focal_lengths, np.float32([2560.56, 2560.56]).astype(np.float64))
    assert len(ws.getCameras()) == 3
    with pytest.raises(KeyError):
        ws.getCamera("missing.jpg")                                 # (the reference prints and exits, Colmap.cpp:60-61)
    with pytest.raises(IndexError):
        ws.getCamera(3)


@pytest.mark.parametrize("name", ["tetra_ascii.ply", "tetra_binary_le.ply", "tetra_binary_be.ply"])
def test_ply_fixture(name, tmp_path):
    mesh = sm.data.Ply(os.path.join(GOLDEN, "ply", name))
    assert mesh.vertices.dtype == np.float32 and mesh.faces.dtype == np.int32
    np.testing.assert_array_equal(mesh.vertices, [[0, 0, 0], [1, 0, 0], [0, 1, 0], [0, 0, 1]])   # colour / quality columns skipped
    np.testing.assert_array_equal(mesh.faces, [[0, 2, 1], [0, 1, 3], [1, 2, 3], [2, 0, 3]])      # the edge element ignored
    # coloured write (python/.../Ply.h:17-36): per-face uint8 red/green/blue; checked by parsing the bytes here, not by Ply()
    colors = np.array([[255, 0, 0], [0, 255, 0], [0, 0, 255], [10, 20, 30]], np.uint8)
    out = str(tmp_path / "out.ply")
    mesh.save(out, colors, False)
    text = open(out).read().split("end_header\n")
    head, body = text[0].split("\n"), text[1].split("\n")
    assert head[0] == "ply" and head[1].startswith("format ascii 1.0")
    assert "element vertex 4" in head and "element face 4" in head
    assert [ln.split()[-1] for ln in head if ln.startswith("property") and "list" not in ln] == ["x", "y", "z", "red", "green", "blue"]
    assert [ln for ln in head if "list" in ln][0].split()[-1] == "vertex_indices"
    assert [float(v) for v in body[1].split()] == [1.0, 0.0, 0.0]
    assert [int(v) for v in body[4 + 3].split()] == [3, 2, 0, 3, 10, 20, 30]
    mesh.save(str(tmp_path / "out_bin.ply"), colors)                 # binary by default
    raw = open(str(tmp_path / "out_bin.ply"), "rb").read()
    payload = raw[raw.index(b"end_header\n") + 11:]
    assert len(payload) == 4 * 12 + 4 * (1 + 12 + 3)
    assert payload[48] == 3 and np.frombuffer(payload[49:61], "<i4").tolist() == [0, 2, 1] and list(payload[61:64]) == [255, 0, 0]


def test_unsupported_camera_model_only_fails_when_used(tmp_path):
    ws = tmp_path
    (ws / "cameras.txt").write_text("1 SIMPLE_RADIAL 640 480 500 320 240 0.01\n2 PINHOLE 640 480 500 501 320 240\n")
    (ws / "images.txt").write_text("1 1 0 0 0 0 0 1 1 a.png\n\n2 1 0 0 0 0 0 1 2 b.png\n\n")
    c = sm.data.Colmap(str(ws))
    assert c.getCamera("b.png").resolution == (640, 480)
    with pytest.raises(ValueError):
        c.getCamera("a.png")
    with pytest.raises(ValueError):
        c.getCameras()


@pytest.mark.gpu
def test_render_from_a_loaded_workspace(oracle):
    """Colmap workspace fixture -> getCamera(file name) -> render a PLY fixture mesh: the path a user of the reference takes
    (README workflow: data.Colmap, data.Ply, render.triangles, renderer.render(camera)); checked against the oracle."""
    from semantic_meshes_amd import _lib
    if _lib.device_count() < 1:
        pytest.fail("gpu test selected but no HIP device is visible")
    ws = sm.data.Colmap(os.path.join(GOLDEN, "colmap_bin"))
    tetra = sm.data.Ply(os.path.join(GOLDEN, "ply", "tetra_binary_le.ply"))
    for name in ("P1180141.JPG", "A0000007 left.png"):
        cam = ws.getCamera(name)
        # put the unit tetrahedron (scaled) 4 units in front of the camera: X_world = R^T (X_cam - t)
        R, t = cam.rotation.astype(np.float64), cam.translation.astype(np.float64)
        pts_cam = tetra.vertices.astype(np.float64) * 1.5 + np.array([-0.4, -0.3, 4.0])
        verts = ((pts_cam - t) @ R).astype(np.float32)            # row vectors: (R^T (x - t))^T = (x - t)^T R
        mesh = sm.data.Mesh(verts, tetra.faces)
        r = sm.render.triangles(mesh)
        idx, depth = r.render(cam)
        o = oracle.OracleRenderer(mesh.vertices, mesh.faces)
        oidx, odepth = o.render(cam)
        idx, depth = np.asarray(idx), np.asarray(depth)
        assert idx.shape == (3072, 2304) and (idx != 0xFFFFFFFF).sum() > 100_000
        assert set(np.unique(idx[idx != 0xFFFFFFFF])) <= {0, 1, 2, 3}
        np.testing.assert_array_equal(idx, oidx)
        np.testing.assert_array_equal(depth.view(np.uint32), odepth.view(np.uint32))
        assert 3.5 < depth[np.isfinite(depth)].min() <= depth[np.isfinite(depth)].max() < 6.0
