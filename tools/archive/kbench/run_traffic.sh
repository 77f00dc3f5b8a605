# HBM-side (fabric) traffic of the GEMM kernels: FETCH_SIZE and WRITE_SIZE in their OWN rocprofv3 passes (kernel-trace only), per MI355X_MICROARCH.md.
# usage: run_traffic.sh  -> gpurun_out/traffic_{fetch,write}_<layout>.md for the w2 layer at 16384 tokens (fwd nt v19, dgrad nn v15, wgrad tn v17)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
run() {  # layout m n k variant
  for C in FETCH_SIZE WRITE_SIZE; do
    rm -rf /tmp/tr_$C
    timeout 200 rocprofv3 --pmc $C --kernel-trace -d /tmp/tr_$C -o r -- tools/kbench/kbench gemm --m $2 --n $3 --k $4 --layout $1 --variants $5 --iters 3 > gpurun_out/traffic_${C}_$1.log 2>&1
    DB=$(find /tmp/tr_$C -name "*.db" | head -1)
    python3 tools/rocprof_summary.py "$DB" gpurun_out/traffic_${C}_$1.md "rocprofv3 --pmc $C --kernel-trace -- kbench gemm --m $2 --n $3 --k $4 --layout $1 --variants $5" > /dev/null 2>>gpurun_out/traffic_${C}_$1.log
    grep -h "gemm_dma_k" gpurun_out/traffic_${C}_$1.md | grep "$C" | cut -c1-220
  done
}
run nt 16384 4096 14336 19
run nn 16384 14336 4096 15
run tn 4096 14336 16384 17
run nt 16384 28672 4096 19
