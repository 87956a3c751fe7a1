"""Drop-in alias: lets code written against the reference package name (`import semantic_meshes`;
`semantic_meshes.data / .render / .fusion`) run on the MI355X implementation in `semantic_meshes_amd`."""
import os as _os
import sys as _sys
import types as _types

import semantic_meshes_amd as _impl
from semantic_meshes_amd import data, fusion  # noqa: F401
from semantic_meshes_amd import render as _render_impl

for _name in ("data", "fusion", "distributed", "synth", "device"):
    _sys.modules[__name__ + "." + _name] = __import__("semantic_meshes_amd." + _name, fromlist=[_name])

# Under the REFERENCE's package name render() returns what the reference returns: a tuple of "dltensor" PyCapsules
# (/root/reference/python/semantic_meshes/include/Renderer.h:37-38), which `tf.experimental.dlpack.from_dlpack`
# (eval-scannet/eval_scannet.py:211-212) requires and `MeshAggregator.add` takes back unconsumed
# (python/scripts/colorize_cityscapes_mesh.py:65-67).  That is a property of the renderers made HERE -- `semantic_meshes.render` is a
# module of its own whose factories pass `capsules=True` -- and not a switch on the shared `semantic_meshes_amd.render` module: a
# process that imports both names keeps getting DeviceArrays (which also speak `__dlpack__` / `__cuda_array_interface__` / numpy) from
# `semantic_meshes_amd.render`, whatever the import order.  SMESH_RENDER_CAPSULES=0 keeps DeviceArrays here too.
_CAPSULES = _os.environ.get("SMESH_RENDER_CAPSULES", "1") != "0"

render = _types.ModuleType(__name__ + ".render", _render_impl.__doc__)
for _k, _v in vars(_render_impl).items():
    if not _k.startswith("__"):
        setattr(render, _k, _v)


def _triangles(mesh, device=0, capsules=None):
    return _render_impl.triangles(mesh, device=device, capsules=_CAPSULES if capsules is None else capsules)


def _texels(mesh, cameras, texels_per_pixel=0.1, device=0, capsules=None):
    return _render_impl.texels(mesh, cameras, texels_per_pixel, device=device, capsules=_CAPSULES if capsules is None else capsules)


_triangles.__doc__, _texels.__doc__ = _render_impl.triangles.__doc__, _render_impl.texels.__doc__
render.triangles, render.texels = _triangles, _texels
_sys.modules[__name__ + ".render"] = render
