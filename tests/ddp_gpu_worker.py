"""Worker of tests/test_gpu_round2.py::test_data_parallel_gradients_equal_mean_of_shard_gradients (2 ranks on one card,
gloo): one forward/backward of DataParallel(AnomalyUNet) on this rank's shard; rank 0 stores the exchanged gradients."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import weights as W  # noqa: E402  (seeded inputs only)

SEED = 1234


def shard_batch(rank):
    image = W.make_input(f"ddp:image{rank}", (2, 3, 32, 48))
    mask = W.make_input(f"ddp:mask{rank}", (2, 1, 32, 48), kind="bernoulli")
    return image, mask


def main():
    import torch.distributed as dist
    import tiaozhanbei_unet_amd as P
    from tiaozhanbei_unet_amd.ddp import DataParallel
    rank = int(os.environ["RANK"])
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    dist.init_process_group("gloo")
    torch.manual_seed(SEED)
    model = P.AnomalyUNet(3, precision="fp32").to(dev).train()
    assert model.two_streams
    net = DataParallel(model, bucket_bytes=16 << 20)
    image, mask = shard_batch(rank)
    recon, amap = net(image.to(dev))
    P.CombinedLoss()(recon, amap, image.to(dev), mask.to(dev))["total_loss"].backward()
    net.finish_gradients()
    torch.cuda.synchronize()
    grads = {k: v.grad.detach().cpu().clone() for k, v in model.named_parameters()}
    flat = torch.cat([g.reshape(-1) for g in grads.values()]).to(dev)
    other = [torch.empty_like(flat) for _ in range(2)]
    dist.all_gather(other, flat)
    same = bool(torch.equal(other[0], other[1]))
    in_place = all(p.grad.data_ptr() == net.exchange._slots[p][1].data_ptr() for p in model.parameters())
    copies_step1 = net.exchange.copies
    # second step on the same batch (no optimiser step in between): the buckets are rebuilt in the completion order of
    # step 1 before this forward; same gradients must come out, again written in place
    for p in model.parameters():
        p.grad = None
    recon, amap = net(image.to(dev))
    P.CombinedLoss()(recon, amap, image.to(dev), mask.to(dev))["total_loss"].backward()
    net.finish_gradients()
    torch.cuda.synchronize()
    step2_same = all(torch.equal(v.grad.cpu(), grads[k]) for k, v in model.named_parameters())
    # ... and on both ranks, with ONE bucket layout (rank 0's completion order was broadcast before the rebuild)
    flat2 = torch.cat([v.grad.reshape(-1) for v in model.parameters()])
    other2 = [torch.empty_like(flat2) for _ in range(2)]
    dist.all_gather(other2, flat2)
    index = {id(p): i for i, p in enumerate(model.parameters())}
    layout = torch.tensor([index[id(p)] for ps in net.exchange._bucket_plan() for p in ps], dtype=torch.int64, device=dev)
    layouts = [torch.empty_like(layout) for _ in range(2)]
    dist.all_gather(layouts, layout)
    ranks_equal2 = bool(torch.equal(other2[0], other2[1])) and bool(torch.equal(layouts[0], layouts[1]))
    if rank == 0:
        torch.save({"grads": grads, "ranks_equal": same, "bucket_mb": net.exchange.bucket_sizes_mb(),
                    "in_place": bool(in_place), "copies": int(net.exchange.copies), "copies_step1": int(copies_step1),
                    "reordered": bool(net.exchange._reordered), "step2_same": bool(step2_same),
                    "ranks_equal_step2": ranks_equal2}, sys.argv[1])
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
