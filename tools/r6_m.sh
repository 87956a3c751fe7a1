#!/bin/bash
root=${GRAFT_REPO_ROOT:-/root/repo}; cd $root; mkdir -p gpurun_out/r6m
( timeout 900 python tools/dump_pixel_stream.py cfg5 8 /tmp/cfg5_stream.bin && timeout 600 tools/stream_bench p /tmp/cfg5_stream.bin ) 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r6m/replay_cfg5.txt
rm -f /tmp/cfg5_stream.bin
