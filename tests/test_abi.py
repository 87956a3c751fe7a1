"""The C-ABI libraries load on a CPU-only box and export every symbol include/smesh.h declares;
the product fails loudly (never falls back) when no GPU is present."""
import ctypes
import os
import re
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "smesh.h")


def declared_symbols():
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(smesh_[a-z0-9_]+)\s*\(", text)))


def test_header_declares_the_boundary():
    syms = declared_symbols()
    for needed in ("smesh_renderer_create_triangles", "smesh_renderer_render", "smesh_aggregator_create",
                   "smesh_aggregator_add", "smesh_aggregator_get", "smesh_aggregator_reset", "smesh_fuse_view"):
        assert needed in syms
    assert len(syms) >= 25


@pytest.fixture(scope="module")
def hip_lib():
    from semantic_meshes_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        subprocess.check_call(["make", "-C", os.path.dirname(_lib.LIB_PATH), "-j4"])
    return ctypes.CDLL(_lib.LIB_PATH)


def test_hip_library_exports_every_declared_symbol(hip_lib):
    missing = [s for s in declared_symbols() if not hasattr(hip_lib, s)]
    assert not missing, missing
    hip_lib.smesh_backend.restype = ctypes.c_char_p
    assert hip_lib.smesh_backend() == b"hip-gfx950"


def test_oracle_library_exports_every_declared_symbol(oracle):
    lib = oracle.lib()
    missing = [s for s in declared_symbols() if not hasattr(lib, s)]
    assert not missing, missing
    assert lib.smesh_backend() == b"oracle-cpu"


def test_product_library_has_no_wrong_result_switches(hip_lib):
    """The development ablations (kernels without their atomics / stores / loads: wrong results, timing only) are compiled out of the
    product build and their environment variables are not read: the names do not even occur in the shared library."""
    from semantic_meshes_amd import _lib
    blob = open(_lib.LIB_PATH, "rb").read()
    for name in (b"SMESH_DBG", b"SMESH_FDBG", b"SMESH_RDBG", b"SMESH_REC_DBG"):
        assert name + b"\0" not in blob and name not in blob, name


def test_python_signature_table_matches_header():
    from semantic_meshes_amd import _lib
    assert sorted(_lib.SIGNATURES) == declared_symbols()


def test_camera_pod_layout_matches_c_struct():
    from semantic_meshes_amd import _lib
    assert ctypes.sizeof(_lib.CameraPOD) == 9 * 4 + 3 * 4 + 2 * 8 + 2 * 8 + 8 + 8 == 96
    assert _lib.CameraPOD.focal.offset == 48 and _lib.CameraPOD.width.offset == 80


def test_product_has_no_cpu_fallback():
    """Without a GPU every compute entry point must raise; with one this test is moot."""
    import semantic_meshes_amd as sm
    from semantic_meshes_amd import _lib
    if _lib.device_count() > 0:
        pytest.skip("a GPU is visible")
    with pytest.raises(RuntimeError):
        sm.fusion.MeshAggregator(10, 5)
    with pytest.raises(RuntimeError):
        sm.render.triangles(sm.data.Mesh(np.zeros((3, 3), np.float32), np.array([[0, 1, 2]], np.int32)))


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "semantic_meshes_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".cpp", ".hpp", ".h")) or f == "Makefile":
                text = open(os.path.join(dirpath, f)).read()
                assert "oracle" not in text.replace("the oracle", "").replace("CPU oracle", "").replace("oracle/", "ORACLEDIR/") or \
                    "import oracle" not in text and "from oracle" not in text and "libsmesh_oracle" not in text, f
                assert "libsmesh_oracle" not in text and "from oracle" not in text and "import oracle" not in text, f


def test_c99_client_compiles_and_runs_against_the_oracle(tmp_path):
    """include/smesh.h is plain C: tests/abi_smoke.c (what a cgo / JNI / C++ host would write) compiles with
    gcc -std=c99 -pedantic, and -- linked against the CPU oracle, which exports the same ABI -- runs."""
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = os.path.join(root, "tests", "abi_smoke.c")
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-I", os.path.join(root, "include"),
                           "-fsyntax-only", src])
    from oracle import oracle
    oracle.build()
    exe = str(tmp_path / "abi_smoke_oracle")
    subprocess.check_call(["gcc", "-std=c99", "-DABI_SMOKE_SKIP_DEVICE_COUNT", "-I", os.path.join(root, "include"), src,
                           "-L", os.path.join(root, "oracle"), "-lsmesh_oracle", "-lm",
                           "-Wl,-rpath," + os.path.join(root, "oracle"), "-o", exe])
    out = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "abi smoke ok" in out.stdout
