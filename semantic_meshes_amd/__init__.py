"""semantic_meshes_amd -- MI355X-native project-and-fuse hot path of semantic-meshes.

Same Python surface as the reference package (/root/reference/python/semantic_meshes/__init__.py:1-4):
sub-modules `data`, `render`, `fusion`; plus `distributed` (view sharding + RCCL sum of the
per-primitive accumulator) and `synth` (benchmark scenes).  Host code is numpy + ctypes over the
C ABI in include/smesh.h; compute is hand-written HIP for gfx950 (csrc/).
"""
from . import data, fusion, render  # noqa: F401

__all__ = ["data", "render", "fusion"]
