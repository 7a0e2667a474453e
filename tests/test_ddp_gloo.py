"""World-size-2 CPU rehearsal (gloo) of the data-parallel gradient exchange: bucket layout, overlap
hooks, averaging and the initial broadcast.  The exchange is model-agnostic, so a small torch model
stands in for the U-Net here (the HIP model itself cannot run without a GPU -- by design)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _net(seed):
    torch.manual_seed(seed)
    return torch.nn.Sequential(torch.nn.Linear(13, 31), torch.nn.Tanh(), torch.nn.Linear(31, 7),
                               torch.nn.Tanh(), torch.nn.Linear(7, 3))


def _worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from tiaozhanbei_unet_amd.ddp import DataParallel
        net = _net(100 + rank)                      # different init per rank: broadcast must fix it
        ddp = DataParallel(net, bucket_bytes=1024)  # tiny buckets => several collectives
        assert len(ddp.exchange.buckets) >= 2
        torch.manual_seed(7)
        data = torch.randn(world, 5, 13)
        target = torch.randn(world, 5, 3)
        losses = []
        for step in range(2):
            net.zero_grad(set_to_none=True)
            loss = ((ddp(data[rank]) - target[rank]) ** 2).mean()
            loss.backward()
            ddp.finish_gradients()
            with torch.no_grad():
                for p in net.parameters():
                    p -= 0.1 * p.grad
            losses.append(float(loss))
        out[rank] = ([p.detach().clone() for p in net.parameters()], [p.grad.clone() for p in net.parameters()])
    finally:
        dist.destroy_process_group()


def test_gradient_exchange_world2_matches_full_batch():
    world, port = 2, _free_port()
    mgr = mp.get_context("spawn").Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(world, port, out), nprocs=world, join=True)
    (p0, g0), (p1, g1) = out[0], out[1]
    for a, b in zip(p0, p1):
        assert torch.equal(a, b), "replicas diverged"
    for a, b in zip(g0, g1):
        assert torch.equal(a, b), "reduced gradients differ between ranks"
    # single-process reference: the mean over ranks of per-rank mean losses == full-batch mean
    net = _net(100)
    torch.manual_seed(7)
    data, target = torch.randn(world, 5, 13), torch.randn(world, 5, 3)
    for step in range(2):
        net.zero_grad(set_to_none=True)
        loss = sum(((net(data[r]) - target[r]) ** 2).mean() for r in range(world)) / world
        loss.backward()
        with torch.no_grad():
            for p in net.parameters():
                p -= 0.1 * p.grad
    for a, b in zip(p0, net.parameters()):
        assert torch.allclose(a, b.detach(), atol=1e-6), "data-parallel result differs from the full-batch step"


def test_bucket_layout_reverse_order_and_alignment():
    from tiaozhanbei_unet_amd.ddp import GradientExchange
    net = _net(0)
    ex = GradientExchange(net.parameters(), bucket_bytes=600)
    params = list(net.parameters())
    assert ex._slots[params[-1]][0] == 0, "last-registered parameter must sit in the first bucket"
    assert ex._slots[params[0]][0] == len(ex.buckets) - 1
    for p in params:
        bi, view = ex._slots[p]
        assert view.shape == p.shape and view.data_ptr() % 16 == 0
    assert sum(b.numel() for b in ex.buckets) >= sum(p.numel() for p in params)


def _subgroup_worker(rank, world, port, out):
    """World of 3; the exchange runs on the sub-group {1, 2}: rank 0 of that group is GLOBAL rank 1."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from tiaozhanbei_unet_amd.ddp import DataParallel
        group = dist.new_group([1, 2])
        if rank == 0:
            out[rank] = "idle"
            dist.barrier()                         # (rank 0 hosts the rendezvous store: it must outlive the others' work)
            return
        net = _net(200 + rank)
        ddp = DataParallel(net, bucket_bytes=1024, process_group=group)
        torch.manual_seed(9)
        data, target = torch.randn(world, 5, 13), torch.randn(world, 5, 3)
        for step in range(2):                      # step 0 ends with the bucket-order broadcast from the group's rank 0
            net.zero_grad(set_to_none=True)
            ((ddp(data[rank]) - target[rank]) ** 2).mean().backward()
            ddp.finish_gradients()
            with torch.no_grad():
                for p in net.parameters():
                    p -= 0.1 * p.grad
        out[rank] = [p.detach().clone() for p in net.parameters()]
        dist.barrier()
    finally:
        dist.destroy_process_group()


def test_exchange_on_a_subgroup_without_global_rank_zero():
    world, port = 3, _free_port()
    mgr = mp.get_context("spawn").Manager()
    out = mgr.dict()
    mp.spawn(_subgroup_worker, args=(world, port, out), nprocs=world, join=True)
    assert out[0] == "idle"
    for a, b in zip(out[1], out[2]):
        assert torch.equal(a, b), "replicas of the sub-group diverged"
    ref = _net(201)                                # the group's rank 0 (global rank 1) seeds both replicas
    torch.manual_seed(9)
    data, target = torch.randn(world, 5, 13), torch.randn(world, 5, 3)
    for step in range(2):
        ref.zero_grad(set_to_none=True)
        (sum(((ref(data[r]) - target[r]) ** 2).mean() for r in (1, 2)) / 2).backward()
        with torch.no_grad():
            for p in ref.parameters():
                p -= 0.1 * p.grad
    for a, b in zip(out[1], ref.parameters()):
        assert torch.allclose(a, b.detach(), atol=1e-6)


def _bf16_worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from tiaozhanbei_unet_amd.ddp import DataParallel
        net = _net(300)
        ddp = DataParallel(net, bucket_bytes=1024, comm_dtype=torch.bfloat16)
        torch.manual_seed(11)
        data, target = torch.randn(world, 5, 13), torch.randn(world, 5, 3)
        net.zero_grad(set_to_none=True)
        ((ddp(data[rank]) - target[rank]) ** 2).mean().backward()
        local = [p.grad.clone() for p in net.parameters()]
        ddp.finish_gradients()
        out[rank] = (local, [p.grad.clone() for p in net.parameters()])
    finally:
        dist.destroy_process_group()


def test_bf16_gradient_buckets_average_the_bf16_rounded_gradients():
    """comm_dtype=bfloat16: what comes back is the average of the ranks' bf16-ROUNDED gradients (rounded to bf16 again by
    the collective), identical on every rank, in the fp32 buckets the optimiser reads."""
    world, port = 2, _free_port()
    mgr = mp.get_context("spawn").Manager()
    out = mgr.dict()
    mp.spawn(_bf16_worker, args=(world, port, out), nprocs=world, join=True)
    (l0, g0), (l1, g1) = out[0], out[1]
    for a, b, x, y in zip(g0, g1, l0, l1):
        assert a.dtype == torch.float32 and torch.equal(a, b)
        want = ((x.bfloat16().float() + y.bfloat16().float()) / 2)
        assert torch.allclose(a, want, rtol=2 ** -7, atol=1e-7), float((a - want).abs().max())
        assert torch.equal(a, a.bfloat16().float()), "values come out of a bf16 collective"
