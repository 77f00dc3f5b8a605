cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05n
timeout 900 python -m pytest tests/test_ring_attention.py tests/test_kernels_gpu.py -x -q -m gpu -k "ring or flash" -p no:cacheprovider 2>&1 | grep -v "amdgpu.ids\|socket.cpp" | tail -40 > gpurun_out/r05n/ring.log 2>&1
tail -40 gpurun_out/r05n/ring.log
