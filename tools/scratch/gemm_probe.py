import torch, time, json
dev="cuda"
def bench(f, n=10):
    for _ in range(3): f()
    torch.cuda.synchronize(); t=time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter()-t)/n
M,K,N=65536,1024,8192
a=torch.randn(M,K,device=dev); w=torch.randn(N,K,device=dev)
fl=2*M*K*N
t=bench(lambda: a@w.t()); print("fp32 highest", t*1e3, "ms", fl/t/1e12, "TF")
torch.set_float32_matmul_precision("high")
t=bench(lambda: a@w.t()); print("fp32 'high'", t*1e3, "ms", fl/t/1e12, "TF")
torch.set_float32_matmul_precision("highest")
ab,wb=a.bfloat16(),w.bfloat16()
t=bench(lambda: ab@wb.t()); print("bf16->bf16", t*1e3, "ms", fl/t/1e12, "TF")
try:
    t=bench(lambda: torch.mm(ab,wb.t(),out_dtype=torch.float32)); print("bf16->fp32 out_dtype", t*1e3, "ms", fl/t/1e12, "TF")
    ok=True
except Exception as e:
    print("out_dtype failed:", repr(e)[:300]); ok=False
# split precision test
def split(x):
    hi=x.bfloat16(); lo=(x-hi.float()).bfloat16(); return hi,lo
ah,al=split(a); wh,wl=split(w)
if ok:
    def x3():
        return torch.mm(ah,wh.t(),out_dtype=torch.float32)+torch.mm(ah,wl.t(),out_dtype=torch.float32)+torch.mm(al,wh.t(),out_dtype=torch.float32)
    t=bench(x3); print("bf16x3 (3 mm + adds)", t*1e3, "ms", fl/t/1e12, "TF-equiv")
    A3=torch.cat([ah,ah,al],1); W3=torch.cat([wh,wl,wh],1)
    t=bench(lambda: torch.mm(A3,W3.t(),out_dtype=torch.float32)); print("bf16x3 (K-concat)", t*1e3, "ms", fl/t/1e12, "TF-equiv")
    ref=(a[:2048].double()@w.double().t())
    for name,val in (("fp32",a[:2048]@w.t()),("bf16",(ab[:2048]@wb.t()).float()),("x3",torch.mm(A3[:2048],W3.t(),out_dtype=torch.float32))):
        e=(val.double()-ref).abs().max().item()/ref.abs().max().item(); print(name,"max rel err vs fp64:",e)
t=bench(lambda: split(a)); print("split activations", t*1e3,"ms")
a16,w16=a.half(),w.half()
t=bench(lambda: a16@w16.t()); print("fp16->fp16", t*1e3, "ms", fl/t/1e12, "TF")
