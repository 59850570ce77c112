import torch, time
dev=torch.device("cuda",0)
x=torch.rand(2048,50176,device=dev); idx=torch.randperm(2048,device=dev)[:256]
def t(f,n=20):
    f(); torch.cuda.synchronize(); e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1)/n*1e3
print("index_select %.1f us"%t(lambda: x.index_select(0,idx)))
print("x[idx]       %.1f us"%t(lambda: x[idx]))
out=torch.empty(256,50176,device=dev)
print("index_select out= %.1f us"%t(lambda: torch.index_select(x,0,idx,out=out)))
print("gather via take_along_dim %.1f us"%t(lambda: torch.take_along_dim(x, idx[:,None].expand(-1,50176), 0)))
print("contiguous slice copy %.1f us"%t(lambda: out.copy_(x[:256])))
