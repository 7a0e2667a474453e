"""How long does the host take to ENQUEUE one training step (no sync)?  If this approaches the step time the
run is launch-bound and graph capture would pay."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tiaozhanbei_unet_amd as P
dev = torch.device("cuda:0")
torch.manual_seed(0)
model = P.AnomalyUNet(3, precision="bf16").to(dev).train()
crit = P.CombinedLoss()
opt = torch.optim.Adam(model.parameters(), lr=1e-3, weight_decay=1e-4, fused=True)
x = torch.randn(32, 3, 256, 256, device=dev)
m = (torch.rand(32, 1, 256, 256, device=dev) < 0.02).float()
def step():
    r, a = model(x)
    l = crit(r, a, x, m)
    opt.zero_grad(set_to_none=True)
    l["total_loss"].backward()
    opt.step()
for _ in range(3): step()
torch.cuda.synchronize()
ts = []
for _ in range(5):
    torch.cuda.synchronize()
    t0 = time.perf_counter(); step(); t1 = time.perf_counter()
    torch.cuda.synchronize(); t2 = time.perf_counter()
    ts.append((t1 - t0, t2 - t0))
for a, b in ts: print(f"enqueue {a*1e3:.2f} ms   step {b*1e3:.2f} ms")
