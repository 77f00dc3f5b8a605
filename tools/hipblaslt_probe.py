import torch, json
dev=torch.device("cuda:0"); bf=torch.bfloat16
T,F,V=4096,14336,92544
def t(fn,it=10):
    fn(); fn(); torch.cuda.synchronize()
    s,e=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(it): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e)*1e-3/it
for name,N,Kd in [("wqkv",6144,4096),("wo",4096,4096),("w13",2*F,4096),("w2",4096,F),("head",V,4096)]:
    X=torch.randn(T,Kd,device=dev).to(bf); W=torch.randn(N,Kd,device=dev).to(bf); DY=torch.randn(T,N,device=dev).to(bf)
    DW=torch.zeros(N,Kd,device=dev,dtype=bf); Y=torch.empty(T,N,device=dev,dtype=bf); DX=torch.empty(T,Kd,device=dev,dtype=bf)
    fl=2.0*T*N*Kd
    r={"gemm":name}
    r["fwd_TF"]=fl/t(lambda: torch.matmul(X,W.t(),out=Y))/1e12
    r["dgrad_TF"]=fl/t(lambda: torch.matmul(DY,W,out=DX))/1e12
    r["wgrad_addmm_TF"]=fl/t(lambda: DW.addmm_(DY.t(),X))/1e12
    print(json.dumps(r),flush=True)
    del X,W,DY,DW,Y,DX
