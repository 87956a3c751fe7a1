"""Drop-in alias: lets code written against the reference package name (`import semantic_meshes`;
`semantic_meshes.data / .render / .fusion`) run on the MI355X implementation in `semantic_meshes_amd`."""
import sys as _sys

import semantic_meshes_amd as _impl
from semantic_meshes_amd import data, fusion, render  # noqa: F401

for _name in ("data", "render", "fusion", "distributed", "synth", "device"):
    _sys.modules[__name__ + "." + _name] = __import__("semantic_meshes_amd." + _name, fromlist=[_name])
