"""Drop-in alias: lets code written against the reference package name (`import semantic_meshes`;
`semantic_meshes.data / .render / .fusion`) run on the MI355X implementation in `semantic_meshes_amd`."""
import sys as _sys

import semantic_meshes_amd as _impl
from semantic_meshes_amd import data, fusion, render  # noqa: F401

for _name in ("data", "render", "fusion", "distributed", "synth", "device"):
    _sys.modules[__name__ + "." + _name] = __import__("semantic_meshes_amd." + _name, fromlist=[_name])

# Under the REFERENCE's package name render() returns what the reference returns: a tuple of "dltensor" PyCapsules
# (/root/reference/python/semantic_meshes/include/Renderer.h:37-38), which `tf.experimental.dlpack.from_dlpack`
# (eval-scannet/eval_scannet.py:211-212) requires and `MeshAggregator.add` takes back unconsumed
# (python/scripts/colorize_cityscapes_mesh.py:65-67).  `semantic_meshes_amd` itself hands out DeviceArrays (which also
# speak `__dlpack__` / `__cuda_array_interface__` / numpy); SMESH_RENDER_CAPSULES=0 keeps those here too.
import os as _os

if _os.environ.get("SMESH_RENDER_CAPSULES", "1") != "0":
    render.RETURN_CAPSULES = True
