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


# ---- rank-divergent parameter use (adapters), frozen parameters and buffers, an aborted backward, the logs -----------
class AdapterNet(nn.Module):
    """a shared trunk + one adapter per 'language': a rank only runs the adapter of its own batch
    (pasero/models/adapters.py:232-301 — why AdapterTransformer asks for find_unused_parameters)"""

    def __init__(self):
        super().__init__()
        self.trunk = nn.Linear(8, 8)
        self.adapters = nn.ModuleDict({k: nn.Linear(8, 8) for k in ('de', 'fr', 'never')})
        self.frozen = nn.Linear(8, 8)
        self.frozen.requires_grad_(False)
        self.register_buffer('table', torch.randn(5, 3))

    def forward(self, x, lang):
        return self.adapters[lang](torch.relu(self.frozen(self.trunk(x)))).pow(2).sum()


def _worker_unused(rank, world, port, ret):
    sys.path.insert(0, ROOT)
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from pasero_amd.ddp import DistributedDataParallel, reduce_logs
    out = {}
    torch.manual_seed(100 + rank)
    net = AdapterNet()
    ddp = DistributedDataParallel(net, bucket_cap_mb=0.0002, find_unused_parameters=True)
    assert len(ddp._buckets) >= 4
    # construction: rank 0's frozen parameters and buffers arrive too
    for name, t in list(net.named_parameters()) + list(net.named_buffers()):
        t0 = t.detach().clone()
        dist.broadcast(t0, 0)
        out[f'bcast:{name}'] = torch.equal(t0, t.detach())
    torch.manual_seed(7)
    full = torch.randn(4, 8)
    mine = full[2 * rank: 2 * rank + 2]
    lang = ('de', 'fr')[rank]
    ref = AdapterNet()
    ref.load_state_dict(net.state_dict())
    (ref(full[:2], 'de') + ref(full[2:], 'fr')).backward()
    ddp(mine, lang).backward()
    for (n, p), (_, q) in zip(net.named_parameters(), ref.named_parameters()):
        if not p.requires_grad:
            continue
        if n.startswith('adapters.never'):
            out[n] = p.grad is None  # unused on every rank: untouched, the optimizer skips it
        else:
            out[n] = p.grad is not None and torch.allclose(p.grad, q.grad / world, rtol=1e-5, atol=1e-6)
    # a backward that dies half-way (the Trainer's OOM path) must not poison the next step
    for p in net.parameters():
        p.grad = None

    class Boom(torch.autograd.Function):
        @staticmethod
        def forward(ctx, x):
            return x.clone()

        @staticmethod
        def backward(ctx, g):
            raise RuntimeError('out of memory (simulated)')

    try:
        h = net.adapters[lang](Boom.apply(torch.relu(net.frozen(net.trunk(mine)))))
        h.pow(2).sum().backward()
        out['boom_raised'] = False
    except RuntimeError:
        out['boom_raised'] = True
    for p in net.parameters():
        p.grad = None
    ddp(mine, lang).backward()
    out['after_abort'] = torch.allclose(net.trunk.weight.grad, ref.trunk.weight.grad / world, rtol=1e-5, atol=1e-6)

    # the training logs: one all-reduce in place of utils.gather_dict
    class Status:
        def __init__(self, value):
            self.value = value

    st = Status(1 if rank == 0 else 3)
    logs = reduce_logs({'loss': 1.5 + rank, 'nll_loss': 1.0 + rank, 'num_tokens': 100 + rank, 'num_lines': 4,
                        'status': st})
    out['logs'] = (abs(logs['loss'] - 4.0) < 1e-12 and abs(logs['nll_loss'] - 3.0) < 1e-12
                   and logs['num_tokens'] == 201 and isinstance(logs['num_tokens'], int) and logs['num_lines'] == 8
                   and logs['status'] is st and st.value == 3)
    # a rank without a batch passes {} (training.py:536-537)
    logs2 = reduce_logs({} if rank == 1 else {'loss': 2.0, 'nll_loss': 1.0, 'num_tokens': 7, 'num_lines': 1})
    out['logs_ragged'] = logs2['num_tokens'] == 7 and abs(logs2['loss'] - 2.0) < 1e-12 and 'status' not in logs2
    try:
        reduce_logs({'loss': 1.0, 'moe_aux': 2.0})
        out['logs_refuse'] = False
    except KeyError:
        out['logs_refuse'] = True
    ret[rank] = out
    dist.barrier()
    dist.destroy_process_group()


def _worker_order(rank, world, port, ret):
    """gradients arrive in a different order on each rank: the collectives must still be issued in bucket order"""
    sys.path.insert(0, ROOT)
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from pasero_amd.ddp import DistributedDataParallel

    class Two(nn.Module):
        def __init__(self):
            super().__init__()
            self.a = nn.Linear(6, 6)
            self.b = nn.Linear(6, 6)

        def forward(self, x, first):
            # two independent branches: which gradient is ready first depends on the graph order, i.e. on the rank
            ya, yb = self.a(x).pow(2).sum(), self.b(x).pow(3).sum()
            return (ya + yb) if first else (yb + ya)

    torch.manual_seed(3)
    net = Two()
    ddp = DistributedDataParallel(net, bucket_cap_mb=0.0001)
    order = []
    orig = ddp._all_reduce_avg
    ddp._all_reduce_avg = lambda b: (order.append(ddp._buckets.index(b)), orig(b))[1]
    x = torch.randn(3, 6, generator=torch.Generator().manual_seed(rank))
    ddp(x, rank == 0).backward()
    ref = Two()
    ref.load_state_dict(net.state_dict())
    xs = [torch.randn(3, 6, generator=torch.Generator().manual_seed(r)) for r in range(world)]
    sum(ref(xx, True) for xx in xs).backward()
    ret[rank] = {'in_order': order == sorted(order) and len(order) == len(ddp._buckets),
                 'grads': all(torch.allclose(p.grad, q.grad / world, rtol=1e-5, atol=1e-6)
                              for p, q in zip(net.parameters(), ref.parameters()))}
    dist.barrier()
    dist.destroy_process_group()


def _run(worker, world=2, timeout=120):
    port = _free_port()
    mgr = mp.Manager()
    ret = mgr.dict()
    ctx = mp.get_context('spawn')
    procs = [ctx.Process(target=worker, args=(r, world, port, ret)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout)
        assert p.exitcode == 0, f'worker exited with {p.exitcode}'
    for r in range(world):
        bad = [k for k, v in ret[r].items() if not v]
        assert not bad, f'rank {r}: {bad}'


def test_ddp_rank_divergent_adapters_frozen_state_abort_and_logs():
    _run(_worker_unused)


def test_ddp_collectives_are_issued_in_bucket_order():
    _run(_worker_order)


def _worker_accum_skip(rank, world, port, ret):
    """update_freq > 1 with adapters: micro-batch 1 (under no_sync) uses adapter 'de', micro-batch 2 (synchronised) does
    not touch it — the accumulated gradient of 'de' is reduced with the rest and must still be there afterwards
    (pasero/training.py:392-408; torch DDP keeps it)"""
    sys.path.insert(0, ROOT)
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from pasero_amd.ddp import DistributedDataParallel
    out = {}
    for find_unused in (False, True):
        torch.manual_seed(5)
        net = AdapterNet()
        ddp = DistributedDataParallel(net, bucket_cap_mb=0.0002, find_unused_parameters=find_unused)
        xs = [torch.randn(2, 8, generator=torch.Generator().manual_seed(10 * r + i)) for r in range(world) for i in (0, 1)]
        x1, x2 = xs[2 * rank], xs[2 * rank + 1]
        ref = AdapterNet()
        ref.load_state_dict(net.state_dict())
        sum(ref(xs[2 * r], 'de') + ref(xs[2 * r + 1], 'fr') for r in range(world)).backward()
        with ddp.no_sync():
            ddp(x1, 'de').backward()
        ddp(x2, 'fr').backward()
        tag = f'unused={find_unused}'
        for (n, p), (_, q) in zip(net.named_parameters(), ref.named_parameters()):
            if not p.requires_grad:
                continue
            if n.startswith('adapters.never'):
                out[f'{tag}:{n}'] = p.grad is None or float(p.grad.abs().max()) == 0.0
            else:
                out[f'{tag}:{n}'] = p.grad is not None and torch.allclose(p.grad, q.grad / world, rtol=1e-5, atol=1e-6)
    ret[rank] = out
    dist.barrier()
    dist.destroy_process_group()


def test_ddp_no_sync_gradient_survives_a_last_micro_batch_that_skips_its_parameter():
    _run(_worker_accum_skip)


@pytest.mark.parametrize('n', [2, 4, 8])
def test_direct_all_reduce_layout_replayed_on_the_host(n):
    """schedule 2 of pk_comm_all_reduce_mean (csrc/comm.hip) for n ranks, replayed with numpy from the offsets the
    library itself computes (pk_comm_direct_plan: host arithmetic, no GPU): exchange 1 into scratch, the fixed-order
    mean of shard_mean_kernel (part r from scratch + r * shard, the rank's own part from the bucket), exchange 2 —
    every rank must end with the mean of the n buckets, and only multiples of 8 n elements are accepted"""
    import ctypes
    import numpy as np
    from pasero_amd import lib
    L = lib.load()
    count = 8 * n * 5
    rng = np.random.default_rng(n)
    bufs = [rng.standard_normal(count).astype(np.float32) for _ in range(n)]
    want = np.mean(np.stack(bufs), axis=0, dtype=np.float64)
    plans = []
    for r in range(n):
        shard, own = ctypes.c_longlong(), ctypes.c_longlong()
        send, recv = (ctypes.c_longlong * n)(), (ctypes.c_longlong * n)()
        assert L.pk_comm_direct_plan(count, n, r, ctypes.byref(shard), ctypes.byref(own), send, recv) == 0
        assert shard.value * n == count and shard.value % 8 == 0 and own.value == r * shard.value
        plans.append((shard.value, own.value, list(send), list(recv)))
    sh = plans[0][0]
    scratch = [np.full(count, np.nan, np.float32) for _ in range(n)]
    for r in range(n):      # exchange 1: r sends buf[send_off[p]] to p, which receives it at scratch[recv_off[r]]
        for p in range(n):
            if p != r:
                so, ro = plans[r][2][p], plans[p][3][r]
                scratch[p][ro: ro + sh] = bufs[r][so: so + sh]
    for r in range(n):      # the reduction kernel's indexing
        own = plans[r][1]
        acc = np.zeros(sh, np.float32)
        for part in range(n):
            acc += bufs[r][own: own + sh] if part == r else scratch[r][part * sh: part * sh + sh]
        bufs[r][own: own + sh] = acc * np.float32(1.0 / n)
    for r in range(n):      # exchange 2: r's reduced shard lands at buf[send_off[r]] of every peer
        own = plans[r][1]
        for p in range(n):
            if p != r:
                dst = plans[p][2][r]
                bufs[p][dst: dst + sh] = bufs[r][own: own + sh]
    for r in range(n):
        assert not np.isnan(bufs[r]).any() and np.allclose(bufs[r], want, rtol=1e-5, atol=1e-6), r
    one = ctypes.c_longlong()
    arr = (ctypes.c_longlong * n)()
    assert L.pk_comm_direct_plan(count + 8, n, 0, ctypes.byref(one), ctypes.byref(one), arr, arr) != 0
    assert b'multiple' in L.pk_last_error()


# ---- the gradient-arena layout of natively run layers (ddp._build_buckets + native_layer.grad_buffers) at world 4 and 8 ----
def _worker_arena(rank, world, port, ret):
    """The real model classes (bf16 parameters, on CPU) under the reducer with the arena layout on: each layer's backward is
    stood in for by a Function that takes its gradient buffers from native_layer.grad_buffers — exactly what
    NativeLayerFn.backward does — and fills them with values that depend on (rank, parameter).  After the backward every
    gradient must be the mean over the ranks AND be the bucket view itself; only parameters outside the layers are packed."""
    sys.path.insert(0, ROOT)
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    os.environ['PASERO_NO_NATIVE_LAYER'] = '0'
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from pasero_amd import ddp as D, native_layer as NL
    from pasero_amd.config import TransformerConfig, DistributedConfig, SyntheticTask
    from pasero_amd.transformer import Transformer
    D._ARENA_ON_ANY_DEVICE = True
    torch.manual_seed(5 + rank)
    cfg = TransformerConfig(embed_dim=64, encoder_ffn_dim=128, decoder_ffn_dim=128, encoder_attention_heads=1,
                            decoder_attention_heads=1, encoder_layers=2, decoder_layers=2, dropout=0.0)
    model = Transformer(cfg, DistributedConfig(dp_size=world, dp_rank=rank), SyntheticTask(40)).to(torch.bfloat16)
    ddp = D.DistributedDataParallel(model, bucket_cap_mb=0.01)
    layers = [(l, False) for l in model.encoder.layers] + [(l, True) for l in model.decoder.layers]
    out = {'arena_layers': len(ddp._arena_layers) == 4, 'buckets': len(ddp._buckets) > 2}

    def value(r, k, p):  # the "gradient" rank r produces for the k-th parameter of a layer: exact in bf16
        return float((r + 1) * (k % 7 + 1)) / 8.0

    class StandIn(torch.autograd.Function):
        @staticmethod
        def forward(ctx, x, layer, is_dec, *params):
            ctx.layer, ctx.is_dec, ctx.params = layer, is_dec, params
            return x + 1.0

        @staticmethod
        def backward(ctx, dy):
            layer, is_dec, params = ctx.layer, ctx.is_dec, ctx.params
            d, f = 64, 128
            wd, w2, vec, wrows, vsz = NL.grad_buffers(layer, params, is_dec, d, f, torch.bfloat16, dy.device)
            grads = NL.grads_in_param_order(wd, w2, vec, wrows, vsz, 2 if is_dec else 1, params)
            for k, g in enumerate(grads):
                g.fill_(value(rank, k, params[k]))
            return (dy, None, None, *grads)

    emb = model.encoder.embed_tokens.weight

    def loss_of():
        x = emb[:4].float().sum() * torch.ones(3)            # (a parameter outside the layers: packed by the reducer)
        for layer, is_dec in layers:
            x = StandIn.apply(x, layer, is_dec, *NL.layer_params(layer, is_dec))
        return x.sum()

    def check(tag, factor):
        ok, views = True, True
        for layer, is_dec in layers:
            for k, p in enumerate(NL.layer_params(layer, is_dec)):
                want = factor * sum(value(r, k, p) for r in range(world)) / world
                ok &= p.grad is not None and bool(torch.allclose(p.grad.float(), torch.full_like(p.grad.float(), want),
                                                                  rtol=1e-2, atol=0))
                b, i = ddp._where[p]
                views &= p.grad.data_ptr() == b.view(i).data_ptr()
        out[tag + ':mean'], out[tag + ':views'] = ok, views

    ddp.module.zero_grad(set_to_none=True)
    loss = loss_of()
    ddp._release_arenas()   # (what ddp.forward does; the stand-in graph is not built through it)
    loss.backward()
    check('step1', 1.0)
    in_layers = sum(len(NL.layer_params(l, d_)) for l, d_ in layers)
    out['only_outside_params_packed'] = ddp.packed_copies == 1        # the embedding
    out['embedding_mean'] = bool(torch.allclose(emb.grad[:4].float(), torch.full((4, 64), 3.0), rtol=1e-2))
    # a second step, and a no_sync micro-batch followed by a reducing one (the second finds `.grad` set: fresh tensors,
    # accumulated by autograd INTO the bucket views)
    ddp.module.zero_grad(set_to_none=True)
    ddp._release_arenas()
    loss_of().backward()
    check('step2', 1.0)
    ddp.module.zero_grad(set_to_none=True)
    with ddp.no_sync():
        ddp._release_arenas()
        loss_of().backward()
    ddp._release_arenas()
    loss_of().backward()
    check('accum', 2.0)
    out['in_layers'] = in_layers == 2 * 16 + 2 * 26
    ret[rank] = out
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize('world', [4, 8])
def test_ddp_gradient_arena_layout_world_4_and_8(world):
    """the bucket-view gradients of natively run layers had only ever met one or two ranks"""
    _run(_worker_arena, world=world, timeout=240)


@pytest.mark.parametrize('world', [4, 8])
def test_ddp_bucket_order_world_4_and_8(world):
    _run(_worker_order, world=world, timeout=240)
