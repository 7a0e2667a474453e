#!/usr/bin/env python3
"""How far are bf16-mode gradients of the full-size AnomalyUNet (N=2, 3x256x256) from the fp32 CPU oracle, per
parameter -- with the round-2 BatchNorm fusions on and off (env UNET_FUSE_BN_BWD / UNET_FUSE_BN_HEAD)?"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import unet_oracle as O, weights as W  # noqa: E402
import tiaozhanbei_unet_amd as P  # noqa: E402

DEV = "cuda:0"
state = W.make_state(W.state_spec("anomaly_unet", 3, 1, False), 0)
image = W.make_input("full:image", (2, 3, 256, 256))
mask = W.make_input("full:mask", (2, 1, 256, 256), kind="bernoulli")
ref_file = "/tmp/full_ref.pt"
if os.path.exists(ref_file):
    ref = torch.load(ref_file)
else:
    torch.set_num_threads(16)
    work = {k: (v.clone().requires_grad_(True) if O.is_trainable(k) else v) for k, v in state.items()}
    r, a = O.anomaly_unet_forward(work, image, True)
    O.combined_loss(r, a, image, mask)["total_loss"].backward()
    ref = {k: v.grad for k, v in work.items() if O.is_trainable(k)}
    torch.save(ref, ref_file)
m = P.AnomalyUNet(3, precision=sys.argv[1] if len(sys.argv) > 1 else "bf16")
m.load_state_dict(state)
m = m.to(DEV).train()
recon, amap = m(image.to(DEV))
P.CombinedLoss()(recon, amap, image.to(DEV), mask.to(DEV))["total_loss"].backward()
torch.cuda.synchronize()
errs = {}
for k, p in m.named_parameters():
    g, w = p.grad.double().cpu(), ref[k].double()
    errs[k] = float((g - w).norm() / (w.norm() + 1e-30))
worst = sorted(errs.items(), key=lambda kv: -kv[1])[:8]
print(f"FUSE_BN_BWD={os.environ.get('UNET_FUSE_BN_BWD', '1')} FUSE_BN_HEAD={os.environ.get('UNET_FUSE_BN_HEAD', '1')} "
      f"DGRAD_BN={os.environ.get('UNET_DGRAD_BN', '1')}: median {sorted(errs.values())[len(errs) // 2]:.3f}  worst: "
      + ", ".join(f"{k.replace('maxpool_conv.1.', '').replace('double_conv', 'dc')}={v:.3f}" for k, v in worst))
