cd $GRAFT_REPO_ROOT
O=gpurun_out/r05s; mkdir -p $O
IE_BENCH_BACKEND=gloo timeout 300 python bench.py --gpus 2 --staged-test --config tiny --steps 3 --warmup 1 --no-cpu-baseline > $O/staged2.json 2> $O/staged2.err; echo "rc=$?"; tail -3 $O/staged2.err | cut -c1-300; cut -c1-600 $O/staged2.json
IE_BENCH_BACKEND=gloo timeout 300 python bench.py --gpus 2 --config tiny --steps 3 --warmup 1 --no-cpu-baseline > $O/staged2_nohook.json 2> $O/staged2_nohook.err; echo "rc without --staged-test=$?"; tail -2 $O/staged2_nohook.err | cut -c1-300
timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29611 bench.py --gpus 1 --config tiny --steps 3 --warmup 1 --no-cpu-baseline > $O/launcher1.json 2> $O/launcher1.err; echo "rc launcher=$?"; cut -c1-300 $O/launcher1.json
