import sys, torch, json
sys.path.insert(0, "/root/repo")
from internevo_amd import kernels as K
dev = torch.device("cuda:0"); bf = torch.bfloat16
T, F = 4096, 14336
def t(fn, it=5):
    fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(it): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) * 1e-3 / it
for name, N, Kd in [("wqkv", 6144, 4096), ("wo", 4096, 4096), ("w13", 2 * F, 4096), ("w2", 4096, F)]:
    X = torch.randn(4 * T, Kd, device=dev).to(bf); DY = torch.randn(4 * T, N, device=dev).to(bf); DW = torch.zeros(N, Kd, device=dev, dtype=bf)
    fl = 2.0 * 4 * T * N * Kd
    def four():
        for i in range(4):
            K.gemm(DY[i * T:(i + 1) * T], X[i * T:(i + 1) * T], True, True, DW, i > 0)
    def one():
        K.gemm(DY, X, True, True, DW, False)
    print(json.dumps({"gemm": name, "4x_accumulate_TF": fl / t(four) / 1e12, "one_K16384_TF": fl / t(one) / 1e12}), flush=True)
    del X, DY, DW
