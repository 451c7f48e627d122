"""The gradient reducer on the GPU through RCCL (the native pk_comm_* path over RCCL's C API; 'nccl' backend for the
communicator's bootstrap).  A GPU box has one card, so the process group has one
rank and PASERO_DDP_FORCE_REDUCE keeps the bucket -> all-reduce(AVG) -> communication-stream -> re-pointed `.grad`
path live; with one rank the reduced gradients must equal the plain ones (every reduction of the step is a fixed-order
sum: bitwise).  (The world_size-2 semantics are
covered on CPU with gloo in test_ddp_cpu.py.)"""
import os
import socket

import pytest
import torch
import torch.distributed as dist

from conftest import load_golden
from model_utils import build_model, text_batch

pytestmark = pytest.mark.gpu


def _free_port():
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


def _base_width_model(dtype):
    """two base-width layers (d = 512): the layers whose weight gradients ride in a grouped launch returned by the
    layer's sink node and whose block ends run inside their GEMMs"""
    import paramgen
    from model_utils import load_paramgen
    from pasero_amd.config import TransformerConfig, DistributedConfig, SyntheticTask
    from pasero_amd.transformer import Transformer
    V = 2000
    model = Transformer(TransformerConfig(dropout=0.1, encoder_layers=2, decoder_layers=2), DistributedConfig(),
                        SyntheticTask(V))
    load_paramgen(model, 3)
    batch = {k: torch.from_numpy(v).cuda() for k, v in paramgen.make_text_batch(5, 16, 64, 64, V, ragged=True).items()}
    return model.to(dtype).cuda(), batch


@pytest.mark.timeout(300)
@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16, 'base_width_bf16', 'base_width_bf16_per_op'])
def test_rccl_bucketed_all_reduce_single_rank(monkeypatch, dtype):
    from pasero_amd import native_layer, rng
    from pasero_amd.ddp import DistributedDataParallel
    per_op = dtype == 'base_width_bf16_per_op'
    if per_op:  # the layers dispatched op by op from Python (the native layer call switched off)
        dtype = 'base_width_bf16'
        monkeypatch.setattr(native_layer, '_OFF', True)
    monkeypatch.setenv('MASTER_ADDR', '127.0.0.1')
    monkeypatch.setenv('MASTER_PORT', str(_free_port()))
    monkeypatch.setenv('PASERO_DDP_FORCE_REDUCE', '1')
    dev = torch.device('cuda', 0)
    dist.init_process_group('nccl', rank=0, world_size=1, device_id=dev)
    try:
        if dtype == 'base_width_bf16':
            dtype = torch.bfloat16
            model, batch = _base_width_model(dtype)
            from pasero_amd import functional as F
            launches = []
            real = F.wgrad_group
            monkeypatch.setattr(F, 'wgrad_group', lambda e: (launches.append(len(e)), real(e))[1])
        else:
            launches = None
            g = load_golden('tiny_encdec_post')
            cfg, model = build_model(g, dtype, 'cuda')
            batch = text_batch(g, 'cuda')
        model.train()
        rng.manual_seed(5)
        loss, _ = model(**batch)
        loss.backward()
        plain = {n: p.grad.clone() for n, p in model.named_parameters() if p.grad is not None}
        for p in model.parameters():
            p.grad = None
        ddp = DistributedDataParallel(model, bucket_cap_mb=0.05)
        assert len(ddp._buckets) > 1
        # the collectives run through RCCL's C API (csrc/comm.hip), not torch.distributed: the communicator came up,
        # every schedule reproduced dist.all_reduce on random data, one was adopted
        assert ddp._native is not None, getattr(__import__('pasero_amd.ddp').ddp.RcclComm._instance, 'report', None)
        rep = ddp._native.report
        assert all(rep['correct'].values()) and rep['schedule'] in ddp._native.SCHEDULES.values(), rep
        rng.manual_seed(5)
        loss2, _ = ddp(**batch)
        loss2.backward()
        torch.cuda.synchronize()
        assert loss2.item() == loss.item()
        if launches is not None:  # both steps launched their weight gradients per layer, under the reducer's hooks too
            # (per-op path: one grouped launch per layer from Python; native layer calls launch theirs from C)
            assert launches == ([7, 7, 4, 4] * 2 if per_op else []), launches
        if launches is not None:
            # a natively run layer writes its gradients into its bucket slice itself (pk_layer_bwd's outputs ARE the slices,
            # native_layer.py / ddp._build_buckets): the reducer packed only what lives outside the layers; op by op every
            # gradient is a tensor of its own and is packed
            in_layers = {p for layer in ddp._arena_layers for piece in native_layer.grad_arena_params(
                layer, hasattr(layer, 'encoder_attn')) for p in piece}
            assert len(ddp._arena_layers) == 4 and len(in_layers) == 2 * 16 + 2 * 26
            with_grad = [p for n, p in model.named_parameters() if n in plain]
            expect = len(with_grad) if per_op else len([p for p in with_grad if p not in in_layers])
            assert ddp.packed_copies == expect, (ddp.packed_copies, expect, len(with_grad))
        rtol = 1e-5 if dtype == torch.float32 else 2e-2
        for n, p in model.named_parameters():
            if n in plain:
                bucket, i = ddp._where[p]
                assert p.grad.data_ptr() == bucket.view(i).data_ptr(), f'{n}: .grad is not the bucket view'
                err = (p.grad.float() - plain[n].float()).abs().max()
                assert err <= rtol * plain[n].float().abs().max(), n
        # gradient accumulation: no_sync micro-batch + reducing micro-batch == 2 x the single gradient
        for p in model.parameters():
            p.grad = None
        with ddp.no_sync():
            rng.manual_seed(5)
            ddp(**batch)[0].backward()
        rng.manual_seed(5)
        ddp(**batch)[0].backward()
        torch.cuda.synchronize()
        for n, p in model.named_parameters():
            if n in plain:
                want = 2 * plain[n].float()
                assert (p.grad.float() - want).abs().max() <= rtol * want.abs().max(), n
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_two_forwards_back_propagated_together_under_the_reducer(monkeypatch):
    """l1 = ddp(b1); l2 = ddp(b2); (l1 + l2).backward(): both autograd nodes of a natively run layer reach their backward
    with `.grad is None` (AccumulateGrad runs after both producers).  Only the first may write the layer's bucket slice —
    the second gets tensors of its own — so the result is g1 + g2, not 2 * g2 (the round-3 counter was reset
    by every forward and let both write the same slice)."""
    from pasero_amd import rng
    from pasero_amd.ddp import DistributedDataParallel
    monkeypatch.setenv('MASTER_ADDR', '127.0.0.1')
    monkeypatch.setenv('MASTER_PORT', str(_free_port()))
    monkeypatch.setenv('PASERO_DDP_FORCE_REDUCE', '1')
    dev = torch.device('cuda', 0)
    dist.init_process_group('nccl', rank=0, world_size=1, device_id=dev)
    try:
        model, b1 = _base_width_model(torch.bfloat16)
        b2 = {k: (v.flip(0) if v.dim() else v) for k, v in b1.items()}   # another batch: the sentences in reverse order
        b2 = {k: v.contiguous() for k, v in b2.items()}
        model.train()
        want = {}
        for seed, b in ((5, b1), (6, b2)):
            for p in model.parameters():
                p.grad = None
            rng.manual_seed(seed)
            model(**b)[0].backward()
            for n, p in model.named_parameters():
                if p.grad is not None:
                    want[n] = want.get(n, 0) + p.grad.float()
        for p in model.parameters():
            p.grad = None
        ddp = DistributedDataParallel(model, bucket_cap_mb=0.05)
        assert len(ddp._arena_layers) == 4
        rng.manual_seed(5)
        l1 = ddp(**b1)[0]
        rng.manual_seed(6)
        l2 = ddp(**b2)[0]
        (l1 + l2).backward()
        torch.cuda.synchronize()
        for n, p in model.named_parameters():
            if n in want:
                err = (p.grad.float() - want[n]).abs().max()
                assert err <= 2e-2 * want[n].abs().max(), (n, err.item(), want[n].abs().max().item())
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_c2_step_under_every_all_reduce_schedule_is_bitwise_the_plain_step(monkeypatch):
    """the C2-width step through the reducer with each of the three native schedules pinned
    (ncclAllReduce, reduce-scatter + all-gather, the direct exchange with its fixed-order shard mean) — on one rank the mean
    over the ranks is the identity, so every gradient must equal the plain step's BIT FOR BIT (dropout 0.1 is on: the reducer must not disturb the order the offsets are drawn in).  What an
    8-GPU node then adds is bandwidth, not arithmetic."""
    from pasero_amd import rng
    from pasero_amd.ddp import DistributedDataParallel
    monkeypatch.setenv('MASTER_ADDR', '127.0.0.1')
    monkeypatch.setenv('MASTER_PORT', str(_free_port()))
    monkeypatch.setenv('PASERO_DDP_FORCE_REDUCE', '1')
    dev = torch.device('cuda', 0)
    dist.init_process_group('nccl', rank=0, world_size=1, device_id=dev)
    try:
        model, batch = _base_width_model(torch.bfloat16)
        model.train()
        rng.manual_seed(5)
        model(**batch)[0].backward()
        plain = {n: p.grad.clone() for n, p in model.named_parameters() if p.grad is not None}
        ddp = DistributedDataParallel(model, bucket_cap_mb=0.05)
        assert ddp._native is not None
        for sched in (0, 1, 2):
            ddp._native.schedule = sched
            for p in model.parameters():
                p.grad = None
            rng.manual_seed(5)
            ddp(**batch)[0].backward()
            torch.cuda.synchronize()
            for n, p in model.named_parameters():
                if n in plain:
                    assert torch.equal(p.grad, plain[n]), (sched, n)
    finally:
        dist.destroy_process_group()


# ------------------------------------------------------------------------------------------------------------
# two ranks, the real HIP model: both processes share the box's one card (gloo carries the CUDA buckets)
# ------------------------------------------------------------------------------------------------------------
def _two_rank_worker(rank, world, port, ret, base_width=False):
    import sys
    from conftest import ROOT
    sys.path.insert(0, ROOT)
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        from pasero_amd import rng
        from pasero_amd.ddp import DistributedDataParallel
        if base_width:  # bf16, d = 512, no dropout: the natively run layers write their gradients into their buckets
            def build():
                import paramgen
                from model_utils import load_paramgen
                from pasero_amd.config import TransformerConfig, DistributedConfig, SyntheticTask
                from pasero_amd.transformer import Transformer
                m = Transformer(TransformerConfig(dropout=0.0, encoder_layers=2, decoder_layers=2), DistributedConfig(),
                                SyntheticTask(2000))
                load_paramgen(m, 3)
                return m.to(torch.bfloat16).cuda()
            import paramgen
            model = build()
            full = {k: torch.from_numpy(v).cuda() for k, v in paramgen.make_text_batch(5, 16, 64, 64, 2000, ragged=True).items()}
        else:
            g = load_golden('tiny_encdec_post')
            build = lambda: build_model(g, torch.float32, 'cuda')[1]
            model = build()
            full = text_batch(g, 'cuda')
        model.train()
        B = full['encoder_input'].size(0)
        rows = [b for b in range(B) if b % world == rank]  # rank r takes every world-th sentence
        mine = {k: v[rows].contiguous() for k, v in full.items()}
        ddp = DistributedDataParallel(model, bucket_cap_mb=0.05)
        loss, logs = ddp(**mine)
        loss.backward()
        torch.cuda.synchronize()
        # single-process truth on the full batch: the loss is a SUM over tokens, DDP AVERAGES over ranks
        ref = build()
        ref.train()
        ref_loss, _ = ref(**full)
        ref_loss.backward()
        worst, name = 0.0, None
        for (n, p), (_, q) in zip(model.named_parameters(), ref.named_parameters()):
            want = q.grad.float() / world
            err = ((p.grad.float() - want).abs().max() / want.abs().max().clamp_min(1e-12)).item()
            if want.abs().max().item() > 1e-6 and err > worst:
                worst, name = err, n
        tot = torch.tensor([loss.item()], dtype=torch.float64)
        dist.all_reduce(tot)
        ret[rank] = {'worst': worst, 'name': name, 'loss_sum': tot.item(), 'ref_loss': ref_loss.item(),
                     'buckets': len(ddp._buckets), 'arena_layers': len(ddp._arena_layers), 'packed': ddp.packed_copies,
                     'with_grad': sum(p.grad is not None for p in model.parameters())}
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(600)
@pytest.mark.parametrize('base_width', [False, True])
def test_two_ranks_average_the_gradients_of_the_hip_model(base_width):
    """world_size 2 with the real model: rank r trains on its share of the batch, the bucketed reducer leaves on every
    rank (sum of the per-rank gradients) / 2 = (full-batch gradient) / 2, and the per-rank losses add up to the
    full-batch loss (fp32: 2e-4 relative, summation order only).  base_width: the bf16 base model, whose natively run
    layers write their gradients straight into their bucket slices on both ranks (bf16 round-off: 3e-2 of the largest
    entry, loss 2e-3)"""
    import torch.multiprocessing as mp
    world, port = 2, _free_port()
    mgr = mp.Manager()
    ret = mgr.dict()
    ctx = mp.get_context('spawn')
    procs = [ctx.Process(target=_two_rank_worker, args=(r, world, port, ret, base_width)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(300)
        assert p.exitcode == 0, f'worker exited with {p.exitcode}'
    for r in range(world):
        out = ret[r]
        assert out['buckets'] > 1
        assert out['worst'] <= (3e-2 if base_width else 2e-4), (r, out['name'], out['worst'])
        assert abs(out['loss_sum'] - out['ref_loss']) <= (2e-3 if base_width else 1e-5) * abs(out['ref_loss'])
        if base_width:  # every layer gradient went in place: only the parameters outside the four layers were packed
            assert out['arena_layers'] == 4 and out['packed'] == out['with_grad'] - (2 * 16 + 2 * 26), out


@pytest.mark.timeout(300)
def test_native_comm_schedules_single_rank(monkeypatch):
    """pk_comm_all_reduce_mean with one rank: all three schedules (all-reduce, reduce-scatter + all-gather, the direct
    exchange with its fixed-order shard mean) leave the buffer unchanged, for every dtype and for sizes that are not a
    multiple of the chunk the kernels use; the multi-rank equivalence with dist.all_reduce is what `RcclComm` itself
    checks at construction on whatever set of GPUs it runs on."""
    from pasero_amd.ddp import RcclComm
    monkeypatch.setenv('MASTER_ADDR', '127.0.0.1')
    monkeypatch.setenv('MASTER_PORT', str(_free_port()))
    dev = torch.device('cuda', 0)
    dist.init_process_group('nccl', rank=0, world_size=1, device_id=dev)
    try:
        comm = RcclComm.get(None, dev)
        assert comm is not None, RcclComm._instance.report
        for dtype in (torch.float32, torch.bfloat16, torch.float16):
            for n in (8, 4096 + 8, 1 << 20):
                x = torch.randn(n, device=dev).to(dtype)
                for sched in (0, 1, 2):
                    y = x.clone()
                    comm.all_reduce_mean(y, sched)
                    torch.cuda.synchronize()
                    assert torch.equal(x, y), (dtype, n, sched)
    finally:
        dist.destroy_process_group()
