import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tiaozhanbei_unet_amd.optim import FusedAdam
DEV = torch.device("cuda:0")
mx = lambda a, b: float((a.detach().cpu().double() - b.detach().cpu().double()).abs().max())
torch.manual_seed(1)
shapes = [(1,), (3,), (64, 3, 3, 3), (4097,), (2, 4096), (129, 64, 3, 3)]
ref_p = [torch.randn(s).requires_grad_(True) for s in shapes]
got_p = [p.detach().clone().to(DEV).requires_grad_(True) for p in ref_p]
kw = dict(lr=3e-3, weight_decay=1e-2)
ref = torch.optim.Adam(ref_p, **kw)
got = FusedAdam(got_p, decoupled=False, **kw)
got.grad_scale = 0.5
for step in range(4):
    for r, g in zip(ref_p, got_p):
        grad = torch.randn(r.shape) * (1.0 + step)
        r.grad = grad.clone()
        g.grad = (grad * 2.0).to(DEV)
    ref.step(); got.step()
print("phase1 p", [mx(g, r) for r, g in zip(ref_p, got_p)])
sd = got.state_dict(); rs = ref.state_dict()["state"]
print("phase1 m", [mx(sd["state"][i]["exp_avg"], rs[i]["exp_avg"]) for i in range(6)])
print("phase1 v", [mx(sd["state"][i]["exp_avg_sq"], rs[i]["exp_avg_sq"]) for i in range(6)])
print("steps", [float(sd["state"][i]["step"]) for i in range(6)], [float(rs[i]["step"]) for i in range(6)])
ref2_p = [p.detach().clone().cpu().requires_grad_(True) for p in got_p]
ref2 = torch.optim.Adam(ref2_p, **kw)
ref2.load_state_dict({"state": {k: {n: (t.cpu() if torch.is_tensor(t) else t) for n, t in v.items()} for k, v in sd["state"].items()},
                      "param_groups": [{k: v for k, v in ref2.state_dict()["param_groups"][0].items()}]})
got.grad_scale = 1.0
for r, g in zip(ref2_p, got_p):
    grad = torch.randn(r.shape)
    r.grad, g.grad = grad.clone(), grad.to(DEV)
ref.step()          # the original torch optimiser too (needs its own grads)
ref2.step(); got.step()
print("phase2 p vs ref2", [mx(g, r) for r, g in zip(ref2_p, got_p)])
print("ref2 steps", [float(v["step"]) for v in ref2.state_dict()["state"].values()])
s2 = got.state_dict()["state"]; r2 = ref2.state_dict()["state"]
print("phase2 m", [mx(s2[i]["exp_avg"], r2[i]["exp_avg"]) for i in range(6)])
print("phase2 v", [mx(s2[i]["exp_avg_sq"], r2[i]["exp_avg_sq"]) for i in range(6)])
