"""Multi-process tests of the data-parallel gradient reducer (pasero_amd/ddp.py) on CPU with the gloo backend,
world_size 2.  The HIP model has no CPU path, so a plain torch MLP stands in for it: what is under test is the reducer
(bucketing, hooks, averaging, no_sync accumulation, unused parameters, parameter broadcast), which is model-agnostic."""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp
import torch.nn as nn

from conftest import ROOT


def _free_port():
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


class Net(nn.Module):
    def __init__(self):
        super().__init__()
        self.a = nn.Linear(16, 32)
        self.b = nn.Linear(32, 32)
        self.c = nn.Linear(32, 8)
        self.unused = nn.Linear(4, 4)  # never part of the graph

    def forward(self, x):
        return self.c(torch.relu(self.b(torch.relu(self.a(x))))).pow(2).sum()


def _worker(rank, world, port, ret):
    sys.path.insert(0, ROOT)
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from pasero_amd.ddp import DistributedDataParallel
    torch.manual_seed(100 + rank)  # different init per rank: the constructor must broadcast rank 0's parameters
    net = Net()
    ddp = DistributedDataParallel(net, bucket_cap_mb=0.002)  # tiny cap -> several buckets
    assert len(ddp._buckets) > 2
    torch.manual_seed(7)
    full = torch.randn(8, 16)  # global batch; rank r takes rows [4r, 4r+4)
    mine = full[4 * rank: 4 * rank + 4]
    # reference: single process, same (broadcast) weights, full batch; loss is a SUM so mean-of-ranks = full / world
    ref = Net()
    ref.load_state_dict(net.state_dict())
    ref(full).backward()
    ddp(mine).backward()
    out = {}
    for (n, p), (_, q) in zip(net.named_parameters(), ref.named_parameters()):
        if n.startswith('unused'):
            out[n] = p.grad is None or float(p.grad.abs().max()) == 0.0
        else:
            out[n] = torch.allclose(p.grad, q.grad / world, rtol=1e-5, atol=1e-6)
    # gradient accumulation: first micro-batch under no_sync (local only), second one reduces the SUM of both
    for p in net.parameters():
        p.grad = None
    with ddp.no_sync():
        ddp(mine[:2]).backward()
    local_only = net.a.weight.grad.clone()
    ddp(mine[2:]).backward()
    out['accum'] = torch.allclose(net.a.weight.grad, ref.a.weight.grad / world, rtol=1e-5, atol=1e-6)
    gathered = [torch.zeros_like(local_only) for _ in range(world)]
    dist.all_gather(gathered, local_only)
    out['no_sync_is_local'] = not torch.allclose(gathered[0], gathered[1])
    # a second step after zero_grad(set_to_none=True) (training.py:329)
    for p in net.parameters():
        p.grad = None
    ddp(mine).backward()
    out['second_step'] = torch.allclose(net.c.weight.grad, ref.c.weight.grad / world, rtol=1e-5, atol=1e-6)
    sd0 = [p.detach().clone() for p in net.parameters()]
    for t in sd0:
        dist.broadcast(t, 0)
    out['params_broadcast'] = all(torch.equal(a, b) for a, b in zip(sd0, net.parameters()))
    ret[rank] = out
    dist.barrier()
    dist.destroy_process_group()


def test_ddp_gloo_world2():
    world = 2
    port = _free_port()
    mgr = mp.Manager()
    ret = mgr.dict()
    ctx = mp.get_context('spawn')
    procs = [ctx.Process(target=_worker, args=(r, world, port, ret)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0, f'worker exited with {p.exitcode}'
    for r in range(world):
        bad = [k for k, v in ret[r].items() if not v]
        assert not bad, f'rank {r}: {bad}'
