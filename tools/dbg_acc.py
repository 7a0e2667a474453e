import ctypes as C, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tiaozhanbei_unet_amd import _lib as L, ops
dev = torch.device("cuda:0"); dt = torch.bfloat16
n, ci, co, h, w = 1, 64, 64, 16, 16
lib = L.lib(); st = C.c_void_p(torch.cuda.current_stream().cuda_stream); p = lambda t: C.c_void_p(t.data_ptr())
gy = torch.zeros(n, co, h, w, device=dev).to(dt).contiguous(memory_format=torch.channels_last)
wt = torch.zeros(co, ci, 3, 3, device=dev)
wpd = ops.pack_weight(wt, L.PACK_CONV_DGRAD, ci, co, dt)
base = torch.arange(ci, device=dev, dtype=torch.float32).view(1, ci, 1, 1).expand(n, ci, h, w)
dx = base.to(dt).contiguous(memory_format=torch.channels_last).clone(memory_format=torch.channels_last)
V = ops._views
L.check(lib.unet_conv3x3(1, n, h, w, V([(gy, 0, 0), None]), p(wpd), ci, V([(dx, 0, 0), None]), ci, 1, 1, st), "acc")
torch.cuda.synchronize()
print("pixel(0,0) channels:", dx[0, :, 0, 0].float().tolist())
print("pixel(5,7) channels:", dx[0, :, 5, 7].float().tolist())
gy = base.to(dt).contiguous(memory_format=torch.channels_last).clone(memory_format=torch.channels_last)
wt = torch.zeros(co, ci, 3, 3, device=dev)
for i in range(64): wt[i, i, 1, 1] = 1.0
wpd = ops.pack_weight(wt, L.PACK_CONV_DGRAD, ci, co, dt)
for acc in (0, 1):
    dx = torch.zeros(n, ci, h, w, device=dev).to(dt).contiguous(memory_format=torch.channels_last)
    L.check(lib.unet_conv3x3(1, n, h, w, V([(gy, 0, 0), None]), p(wpd), ci, V([(dx, 0, 0), None]), ci, acc, 1, st), "acc")
    torch.cuda.synchronize()
    print("acc", acc, "pixel(5,7):", dx[0, :, 5, 7].float().tolist())
