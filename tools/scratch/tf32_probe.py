import torch, time, os, sys
sys.path.insert(0, os.getcwd())
dev="cuda"
M,K,N=4096,1024,8192
torch.manual_seed(0)
a=torch.randn(M,K,device=dev); w=torch.randn(N,K,device=dev)
ref=a.double()@w.double().t()
for mode in ("highest","high","medium"):
    torch.set_float32_matmul_precision(mode)
    v=a@w.t()
    e=(v.double()-ref).abs()
    print(mode, "allow_tf32=",torch.backends.cuda.matmul.allow_tf32, "max err/max|ref|", (e.max()/ref.abs().max()).item(), "rms rel", (e.pow(2).mean().sqrt()/ref.pow(2).mean().sqrt()).item())
    v2=torch.nn.functional.linear(a,w)
    print("   F.linear same:", torch.equal(v,v2))
# model level: DiM-L/2 forward, B=8, highest vs high
from bench import build_model
torch.set_float32_matmul_precision("highest")
m=build_model("DiM-L/2", dev)
x=torch.randn(8,4,32,32,device=dev); t=torch.rand(8,device=dev); y=torch.randint(0,1000,(8,),device=dev)
with torch.no_grad():
    o1=m(x,t,y)
    torch.set_float32_matmul_precision("high")
    o2=m(x,t,y)
    m64=None
print("model out: max|o|", o1.abs().max().item(), "max diff high vs highest", (o1-o2).abs().max().item(), "rel", ((o1-o2).abs().max()/o1.abs().max()).item())
