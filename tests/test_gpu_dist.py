"""GPU: the result exchange on the RCCL backend itself (`backend='nccl'` on ROCm), as far as a
one-GPU box allows: a one-rank process group, the collective forced -- RCCL accepts the record
buffer's shape / dtype, the interleave and unpacking are right on device tensors.  World sizes
> 1 are covered on CPU with gloo (tests/test_dist_gloo.py); N > 1 on hardware is the driver's."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.timeout(180)
def test_all_gather_detections_on_rccl_one_rank():
    from iouaware import dist as idist
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(_free_port()), RANK='0',
                      WORLD_SIZE='1', LOCAL_RANK='0')
    rank, world = idist.init_dist('pytorch', backend='nccl')
    try:
        assert (rank, world) == (0, 1) and dist.get_backend() == 'nccl'
        g = torch.Generator(device='cuda').manual_seed(0)
        dets = torch.rand(8, 100, 5, device='cuda', generator=g) * 1000
        labels = torch.randint(0, 80, (8, 100), device='cuda', generator=g, dtype=torch.int32)
        num = torch.randint(0, 101, (8,), device='cuda', generator=g, dtype=torch.int32)
        D, L, N = idist.all_gather_detections(dets, labels, num, force_collective=True)
        torch.cuda.synchronize()
        assert torch.equal(D, dets) and torch.equal(L, labels) and torch.equal(N, num)
        D2, _, _ = idist.all_gather_detections(dets, labels, num, num_samples=5, force_collective=True)
        assert D2.shape[0] == 5
        # the gradient all-reduce of the training path on RCCL (one rank: identity)
        from iouaware.train import allreduce_grads
        net = torch.nn.Linear(7, 3).cuda()
        net(torch.ones(2, 7, device='cuda')).sum().backward()
        before = [p.grad.clone() for p in net.parameters()]
        allreduce_grads(net)
        assert all(torch.equal(a, p.grad) for a, p in zip(before, net.parameters()))
        dist.barrier(device_ids=[0])
    finally:
        dist.destroy_process_group()
