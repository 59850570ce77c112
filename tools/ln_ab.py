import os, sys, torch
sys.path.insert(0, os.getcwd())
import vit_amd.functional as vf
dev = torch.device("cuda:0")
M, D = 50432, 768
g = torch.Generator(device="cpu").manual_seed(0)
dy = torch.randn((M, D), generator=g).to(dev).to(torch.bfloat16)
x = torch.randn((M, D), generator=g).to(dev)
gam = torch.randn(D, generator=g).to(dev)
mean = x.mean(1); rstd = 1.0 / (x.var(1, unbiased=False) + 1e-12).sqrt()
dres = torch.randn((M, D), generator=g).to(dev)
dx = torch.empty_like(x); dgam = torch.empty(D, device=dev); dbet = torch.empty(D, device=dev); dyn = torch.empty((M, D), device=dev, dtype=torch.bfloat16); dbias = torch.empty(D, device=dev)
def run():
    vf.layernorm_bwd_fused(dy, x, gam, mean, rstd, dres, dx, dgam, dbet, dyn, dbias, (0.1, 7, 3))
for _ in range(30): run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(50): run()
e1.record(); torch.cuda.synchronize()
print(os.environ.get("VIT_AMD_LIB", "prod").split("_")[-1], f"ln_bwd_fused + reducers: {e0.elapsed_time(e1) / 50 * 1e3:.1f} us", float(dgam.sum()), float(dbias.sum()))
