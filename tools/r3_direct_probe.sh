cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/probe
SMESH_RASTER=direct timeout 1300 python -m pytest tests/test_gpu_parity.py -q -x -m gpu -k "(render_small_scene or render_cfg1 or overlap or mixed_triangle or texel or fuse_view_cfg2 or triangle_order or fuse_views or near_plane or room) and not (wide_rows_equal or texels_equal)" -p no:cacheprovider --durations=12 -o faulthandler_timeout=400 2>&1 | tail -60 > gpurun_out/probe/direct.log
tail -40 gpurun_out/probe/direct.log
