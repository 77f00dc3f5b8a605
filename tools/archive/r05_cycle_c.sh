# round-5 GPU cycle C: effective shader clock per kernel (GRBM_GUI_ACTIVE / duration) in the training step and in kbench loops -- the evidence behind "the step's
# GEMMs hold the chip at its power limit and the attention kernels inherit their clock" -- and the HBM traffic of the step's GEMMs at HEAD
cd $GRAFT_REPO_ROOT
TAG=${1:-r05c}
O=gpurun_out/$TAG; mkdir -p $O
export TMPDIR=/tmp
K=tools/kbench/kbench
rm -rf /tmp/pm_step; timeout 400 rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace -d /tmp/pm_step -o r -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-timing > $O/bench_under_pmc_clock.json 2> $O/pmc_clock.err
python3 tools/kernel_clock.py "$(find /tmp/pm_step -name '*.db' | head -1)" $O/clock_in_step.md "rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-timing" | head -16
rm -rf /tmp/pm_kb; timeout 200 rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace -d /tmp/pm_kb -o r -- $K fwd --variants 2 --iters 200 > $O/kb_fwd.log 2>&1
python3 tools/kernel_clock.py "$(find /tmp/pm_kb -name '*.db' | head -1)" $O/clock_kbench_fwd.md "rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace -- kbench fwd --variants 2 --iters 200" | grep flash
rm -rf /tmp/pm_kb; timeout 200 rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace -d /tmp/pm_kb -o r -- $K bwd --variants 0 --iters 100 > $O/kb_bwd.log 2>&1
python3 tools/kernel_clock.py "$(find /tmp/pm_kb -name '*.db' | head -1)" $O/clock_kbench_bwd.md "rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace -- kbench bwd --variants 0 --iters 100" | grep flash
rm -rf /tmp/pm_kb; timeout 200 rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace -d /tmp/pm_kb -o r -- $K gemm --m 16384 --n 4096 --k 14336 --layout nt --variants -1 --iters 200 > $O/kb_gemm.log 2>&1
python3 tools/kernel_clock.py "$(find /tmp/pm_kb -name '*.db' | head -1)" $O/clock_kbench_gemm.md "rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace -- kbench gemm nt 16384x4096x14336 --iters 200" | grep gemm
