cd $GRAFT_REPO_ROOT
O=gpurun_out/r05t; mkdir -p $O
IE_TEST_FULL=1 timeout 1500 python -m pytest tests/test_engine_gpu.py -x -q -m gpu -k "live_oracle" -s -p no:xdist 2>&1 | grep -v "amdgpu.ids\|socket.cpp" | tail -30 > $O/live_oracle.log
tail -30 $O/live_oracle.log
