# round-4 GPU cycle B: targeted tests of the changes since cycle A, the bench diagnostics on two staged ranks, yardstick, trace
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04b; mkdir -p $O
export TMPDIR=/tmp IE_TEST_SERIAL=1
( time timeout 900 python -m pytest tests/test_train_entry.py "tests/test_kernels_gpu.py::test_embedding_gradient_of_the_benchmark_batch_against_fp64" -q -m gpu -s -x 2>&1 | grep -v "amdgpu.ids\|socket.cpp" | tail -40 ) > $O/t_entry.log 2>&1; tail -4 $O/t_entry.log
( time timeout 900 python -m pytest tests/test_multirank_gpu.py tests/test_internlm1_gpu.py -q -m gpu -s -k "sequence_sharded or (tensor_and_pipeline and msp)" 2>&1 | grep -v "amdgpu.ids\|socket.cpp" | tail -40 ) > $O/t_ss.log 2>&1; tail -4 $O/t_ss.log
( time timeout 900 python -m pytest "tests/test_engine_gpu.py::test_engine_7b_width_merged_benchmark_step_matches_oracle" -q -m gpu -s 2>&1 | grep -v "amdgpu.ids\|socket.cpp" | tail -60 ) > $O/t_7b.log 2>&1; tail -4 $O/t_7b.log
IE_BENCH_BACKEND=gloo timeout 300 python bench.py --gpus 2 --config tiny --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_staged2.json 2> $O/bench_staged2.err; echo "staged2 rc=$?"; cut -c1-300 $O/bench_staged2.json
IE_BENCH_BACKEND=gloo timeout 300 python bench.py --gpus 2 --config tiny --steps 3 --warmup 1 --no-cpu-baseline --rs-under-w13-only --rccl-channels 4 > $O/bench_staged2_sw.json 2> $O/bench_staged2_sw.err; echo "staged2 switches rc=$?"
timeout 600 python bench.py --steps 10 --warmup 3 > $O/bench_line.json 2> $O/bench_err.log; echo "bench rc=$?"; cut -c1-400 $O/bench_line.json
timeout 300 python tools/hipblaslt_probe.py --tokens 16384 > $O/hipblaslt.log 2>&1; tail -12 $O/hipblaslt.log
rm -rf /tmp/prof_b
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_b -o bench -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline > $O/bench_under_rocprof.json 2> $O/rocprof_err.log
DB=$(find /tmp/prof_b -name "*.db" | head -1)
python3 tools/rocprof_summary.py "$DB" $O/kernel_stats.md "rocprofv3 --kernel-trace --stats -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline" | head -14
python3 tools/gpu_idle_from_trace.py "$DB" > $O/gpu_idle.md 2>&1; cat $O/gpu_idle.md
