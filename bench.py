#!/usr/bin/env python3
"""Benchmark of the Pasero Transformer training hot path on MI355X (BASELINE.json metric).

  python bench.py [--gpus N] [--steps K] [--warmup W]
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
         bench.py --gpus N --steps K --warmup W

`python bench.py --gpus N` with N > 1 and no launcher environment starts the N ranks itself (one process per GPU, like
the reference's own `pasero-train`, cli/train.py:705-727): before anything touches the GPU the parent spawns
`python -m torch.distributed.run ... bench.py <same arguments>` as a child, passes its output through and exits with
its return code.

One "step" = one pass of the hot path over one synthetic batch per GPU: Transformer.forward (encoder, decoder, fused
tied-projection + label-smoothed CE, the logs' device->host copy) + loss.backward() + (N > 1) the bucketed gradient
all-reduce over RCCL, i.e. what `Trainer.train_step` does between `zero_grad` and the optimizer (pasero/training.py:
329-408).  Workload: BASELINE configs[1] — `transformer` base (6+6, d=512, 8 heads, ffn 2048, V=8032), bf16, batch
(256, 128, 128) per GPU, dropout 0.1, label smoothing 0.1, inputs resident in HBM before the timed region.
Metric: target tokens/s (non-pad positions of decoder_input[:, 1:], the reference's own `wps`), whole job.

Rank 0 prints ONE JSON line (schema in the task contract) with two extra objects:
  roofline     — the dominant kernel (bf16 MFMA GEMM instantiation with the largest total time): algorithmic FLOPs per
                 launch (2*M*N*K) / average launch duration, measured with HIP events on the launch stream INSIDE the
                 timed region, against the 2.5 PFLOP/s dense bf16 MFMA peak;
  cpu_baseline — the CPU oracle (oracle/ref_cpu.py, a port pinned to the reference by golden vectors) timed on the
                 host cores on BASELINE configs[0] (8 x (64, 64), fp32), a bounded ~10-30 s sample.
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests', 'golden'))

PEAK_BF16_TFLOPS = 2500.0  # /opt/skills/guides/MI355X_MICROARCH.md: ~2.5 PFLOP/s dense bf16 MFMA
PEAK_HBM_GBPS = 8000.0     # same guide: 8 TB/s HBM3E peak (~6.3 TB/s achievable)

WORKLOADS = {
    # name: (config class name, V, B, S, T)
    'c2_base_bf16': ('TransformerConfig', 8032, 256, 128, 128),
    'c1_base': ('TransformerConfig', 8032, 8, 64, 64),
    'c3_big': ('TransformerBigConfig', 70376, 256, 128, 128),
    'c5_nllb_1b3': ('NLLB1B3Config', 256206, 64, 128, 128),
    # C4 (speech): 30 s clips -> log-mel on the device (K8) -> conv subsampler (K7) -> whisper_base-shaped 6+6 enc-dec;
    # S = 3000 mel frames in, 1500 encoder positions; the log-mel kernel is inside the timed step
    'c4_whisper': ('WhisperConfig', 51865, 16, 3000, 64),
    # C4, second half: the IWSLT2023 recipe as the reference ships it (examples/IWSLT2023/
    # xlsr+nllb-iwslt2021.yaml): (B, S = 1000, 1024) wav2vec-style features -> in_linear 1024 -> 80 + ReLU -> conv k5 s2 + GLU ->
    # 500 positions -> NLLB-1.3B-shaped 24 + 24 with bottleneck adapters on encoder layers 3..23; only in_linear, the
    # subsampler, encoder layers 0-2 and the adapters train (`train_params_regex`, applied as cli/train.py:237-238 does),
    # dropout 0.3, attention dropout 0.1, label smoothing 0.2
    'c4_iwslt': ('AdapterNLLB1B3Config', 256206, 32, 1000, 64),
}
IWSLT_OVERRIDES = dict(input_dim=1024, conv_input_dim=80, conv_kernel_sizes=[5], encoder_positional_encoding='sinusoidal',
                       encoder_embed_norm=False, encoder_adapter_layer_ids=list(range(3, 24)), dropout=0.3,
                       attention_dropout=0.1, label_smoothing=0.2, encoder_max_len=2048, decoder_max_len=128)
IWSLT_TRAIN_REGEX = r'(.*\.in_linear|.*\.subsample|encoder\.layers\.[0-2]\.|.*\.adapters|encoder\.layernorm_embedding)'
SPEECH = ('c4_whisper', 'c4_iwslt')       # S counts input frames; the encoder layers see S / 2 positions
DEVICE_INIT = ('c5_nllb_1b3', 'c4_iwslt')  # 1.4 G parameters: created and drawn on the device (a CPU init takes a minute)


def build_workload(workload: str, dtype, device, rank: int = 0, world: int = 1):
    """model (random-init weights, identical on every rank), configuration and one synthetic batch of `workload` on `device`
    -> (cfg, model, batch, wav); `wav` is the (B, 480 000) fp32 audio of the Whisper workload (its log-mel runs inside the
    step), None otherwise.  Also used by tools/gemm_in_model.py (the GEMM census of a step)."""
    from pasero_amd import config as C, rng
    from pasero_amd import adapters  # noqa: F401  (registers adapter_transformer: the IWSLT recipe's architecture)
    cfg_name, V, B, S, T = WORKLOADS[workload]
    # dropout 0.1, label smoothing 0.1: the training configuration (the IWSLT recipe brings its own)
    cfg = getattr(C, cfg_name)(**(IWSLT_OVERRIDES if workload == 'c4_iwslt' else {}))
    arch = C.get_architecture(cfg)
    dist_cfg = C.DistributedConfig(dp_size=world, dp_rank=rank)
    if workload in DEVICE_INIT:
        from pasero_amd import modules
        with modules.fast_init(device, dtype):
            model = arch(cfg, dist_cfg, C.SyntheticTask(V))
        model = model.to(dtype).to(device)
        gen = torch.Generator(device=device).manual_seed(1234)  # identical random-init weights on every rank
        with torch.no_grad():
            for n, p in model.named_parameters():
                if p.dim() == 1:
                    p.fill_(1.0) if ('norm' in n and n.endswith('weight')) else p.zero_()
                else:
                    p.copy_(torch.randn(p.shape, generator=gen, device=device, dtype=torch.float32) * 0.02)
    else:
        torch.manual_seed(1234)  # identical random-init weights on every rank
        model = arch(cfg, dist_cfg, C.SyntheticTask(V)).to(dtype).to(device)
    if workload == 'c4_iwslt':  # cli/train.py:237-238
        import re
        for n, p in model.named_parameters():
            p.requires_grad = bool(re.match(IWSLT_TRAIN_REGEX, n))
    model.train()
    rng.manual_seed(1 + rank)
    batch = synthetic_batch(B, S if workload not in SPEECH else 4, T, V, seed=1 + rank, device=device)
    wav = None
    if workload == 'c4_whisper':  # wav ~ N(0, 0.1^2) fp32, 30 s at 16 kHz
        gen = torch.Generator().manual_seed(rank)
        wav = (0.1 * torch.randn(B, 480000, generator=gen)).to(device)
        batch['encoder_input_length'] = torch.full((B,), S, dtype=torch.int64, device=device)
    if workload == 'c4_iwslt':    # features ~ N(0, 1), S = 1000 frames of 1024
        gen = torch.Generator().manual_seed(rank)
        batch['encoder_input'] = torch.randn(B, S, cfg.input_dim, generator=gen).to(dtype).to(device)
        batch['encoder_input_length'] = torch.full((B,), S, dtype=torch.int64, device=device)
    return cfg, model, batch, wav


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=100)
    ap.add_argument('--warmup', type=int, default=10)
    ap.add_argument('--workload', default='c2_base_bf16', choices=list(WORKLOADS))
    ap.add_argument('--dtype', default='bf16', choices=['bf16', 'f16', 'f32'])
    ap.add_argument('--force-ddp', action='store_true',
                    help='rehearsal: run the bucketed all-reduce path even on one rank (exercises RCCL on a 1-GPU box)')
    ap.add_argument('--backend', default='nccl', help="torch.distributed backend ('nccl' = RCCL; 'gloo' to rehearse)")
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-roofline', action='store_true', help='skip the per-launch HIP-event instrumentation')
    ap.add_argument('--no-live-traffic', action='store_true',
                    help='roofline.traffic from the committed profiles/ instead of two rocprofv3 counter passes now')
    ap.add_argument('--cpu-seconds', type=float, default=15.0, help='budget of the CPU baseline sample')
    ap.add_argument('--no-extra-workloads', action='store_true',
                    help='skip the c3_big / c4_whisper measurements that follow the headline region at N = 1')
    ap.add_argument('--extra-steps', type=int, default=10, help='timed steps of each extra workload (3 warm-up steps)')
    ap.add_argument('--rehearse-cpu', action='store_true',
                    help='CPU rehearsal of the multi-rank plumbing (launcher, rendezvous, reducer, fused logs '
                         'all-reduce, timing protocol, JSON): a small torch MLP stands in for the HIP model, which has '
                         'no CPU path.  The line it prints is marked "rehearsal" and is not a measurement.')
    return ap.parse_args()


def self_launch(args) -> int:
    """`--gpus N` without a launcher: start N ranks (one process per GPU) as a CHILD torch.distributed.run and return
    its exit code.  Nothing in this process has touched the GPU yet (and nothing is exec'ed)."""
    import socket
    import subprocess
    with socket.socket() as sock:
        sock.bind(('127.0.0.1', 0))
        port = sock.getsockname()[1]
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', f'--nproc-per-node={args.gpus}',
           '--master-addr', '127.0.0.1', '--master-port', str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')  # dmabuf IPC: what RCCL needs on this driver
    env.setdefault('OMP_NUM_THREADS', '4')
    return subprocess.call(cmd, env=env)


class ClockSampler:
    """the shader clock the chip holds during the timed region: the amdgpu hwmon `freq1_input` (Hz) of THIS device (found by
    its PCI address), read every 20 ms by a side thread — boxes of the pool differ by 3-5 % on the same binary, mostly through
    the clock their power envelope settles at.  None where the file is not readable."""

    def __init__(self, device):
        import glob
        self.path, self.samples, self._stop, self._thread = None, [], False, None
        try:
            pr = torch.cuda.get_device_properties(device)
            bdf = f'{pr.pci_domain_id:04x}:{pr.pci_bus_id:02x}:{pr.pci_device_id:02x}.0'
            hits = glob.glob(f'/sys/bus/pci/devices/{bdf}/hwmon/hwmon*/freq1_input')
            self.path = hits[0] if hits else None
        except Exception:
            self.path = None

    def _run(self):
        while not self._stop:
            try:
                with open(self.path) as f:
                    self.samples.append(int(f.read().strip()) / 1e6)
            except Exception:
                break
            time.sleep(0.02)

    def start(self):
        if self.path:
            import threading
            self._thread = threading.Thread(target=self._run, daemon=True)
            self._thread.start()

    def stop(self):
        self._stop = True
        if self._thread is not None:
            self._thread.join(1.0)
        if not self.samples:
            return None
        v = sorted(self.samples)
        return {'min': v[0], 'median': v[len(v) // 2], 'max': v[-1], 'samples': len(v), 'source': 'amdgpu hwmon freq1_input, 20 ms'}


def synthetic_batch(B, S, T, V, seed, device):
    """SURVEY §8d recipe, full-length rows: ids ~ U[4, V), last source token / first+last decoder tokens = 2"""
    import paramgen
    b = paramgen.make_text_batch(seed, B, S, T, V, ragged=False)
    return {k: torch.from_numpy(v).to(device) for k, v in b.items()}


class GemmTimer:
    """per-launch durations of the GEMM kernels, measured by the library itself: HIP events recorded around exactly the
    main GEMM kernel of every STRIDE-th pk_gemm call, on the stream it is launched on (include/pasero_hip.h:
    pk_gemm_timing_*).  Kernel names are the rocprofv3 names, so the averages can be checked against profiles/."""

    STRIDE = 11  # coprime with the 210 GEMM launches of a step (2*3*5*7), so over the steps every launch is sampled

    def __init__(self):
        self.samples = 0
        self.launches_per_step = 0

    def start(self, max_samples: int, stride: int = 0):
        from pasero_amd import lib
        lib.check(lib.load().pk_gemm_timing_start(int(max_samples), stride or self.STRIDE), 'pk_gemm_timing_start')

    def stop(self):
        from pasero_amd import lib
        self.samples = lib.load().pk_gemm_timing_stop()

    @staticmethod
    def kernel_name(kernel, a_col, b_col, dtype):
        tf = {0: 'false', 1: 'true'}
        t = {0: 'float', 1: '__hip_bfloat16', 2: '_Float16'}[dtype]
        if kernel == (8 | 0x40):  # the grouped weight-gradient launch (pk_gemm_wgrad_group): flops = sum over the group
            return 'gemm8p_group_kernel<%s>' % t
        if kernel == (8 | 0x80):  # Linear + residual + dropout + LayerNorm (pk_gemm_ln_fwd)
            return 'gemm8p_ln_kernel<%s, %d>' % (t, a_col)  # (a_col: the epilogue specialisation, see pk_gemm_ln_fwd)
        if kernel == 64:  # the few-rows kernel (gemm_skinny.hip; its activation / mode template arguments are not in the sample)
            return 'gemm_skinny_kernel<%s, ...>' % t
        if kernel & 0x4000:  # the persistent walk of 256 x 256 tiles (round 6): <T, B_COL, BITS>
            return 'gemm8p_pt_kernel<%s, %s, %s>' % (t, tf[b_col], tf[(kernel >> 12) & 1 or (kernel >> 13) & 1])
        if kernel & 0x800:   # gemmpw.hip (off by default): <T, B_COL, EPI>
            return 'gemm8p_pw_kernel<%s, %s, %d>' % (t, tf[b_col], 2 if kernel & 0x2000 else (1 if kernel & 0x1000 else 0))
        if kernel & 0xF == 8 and (kernel & ~0x3400) < 256:
            # gemm8p.hip: 0x10 general epilogue, 0x20 partial last K-tile, 0x400 the 128 x 256 tile (a kernel of its own since
            # round 4: <T, B_COL, TAIL, BITS>), 0x1000 the ReLU mask as bits (round 5; 0x2000: the dH GEMM that reads them);
            # gemm8p_kernel<T, A_COL, B_COL, ANY epilogue, TAIL K-tile, half-M form, BITS> — the names rocprofv3 prints
            bits = tf[(kernel >> 12) & 1]
            if kernel & 0x400:
                return 'gemm8p_hm2_kernel<%s, %s, %s, %s>' % (t, tf[b_col], tf[(kernel >> 5) & 1], bits)
            return 'gemm8p_kernel<%s, %s, %s, %s, %s, false, %s>' % (t, tf[a_col], tf[b_col], tf[(kernel >> 4) & 1],
                                                                       tf[(kernel >> 5) & 1], bits)
        if kernel & 0x200:  # the B-stationary kernel (gemmbs.hip): <T, B_COL, K-tiles, activation, act'-mask, mask as bits, preact>
            return 'gemmbs_kernel<%s, %s, %d, %d, %s, %s, %s>' % (t, tf[b_col], kernel & 0xF, (kernel >> 4) & 3, tf[(kernel >> 6) & 1],
                                                                tf[(kernel >> 7) & 1], tf[(kernel >> 8) & 1])
        if kernel == 256:
            return 'gemm256_kernel<%s, %s, %s, 8>' % (t, tf[a_col], tf[b_col])
        return 'gemm_kernel<%s, %s, %s>' % (t, tf[a_col], tf[b_col])

    def summary(self):
        import ctypes
        from pasero_amd import lib
        L = lib.load()
        ints = [ctypes.c_int() for _ in range(5)]
        flops, ms = ctypes.c_double(), ctypes.c_float()
        agg = {}
        for i in range(self.samples):
            lib.check(L.pk_gemm_timing_read(i, *[ctypes.byref(x) for x in ints], ctypes.byref(flops),
                                            ctypes.byref(ms)), 'pk_gemm_timing_read')
            kernel, a_col, b_col, splitk, dtype = (x.value for x in ints)
            a = agg.setdefault(self.kernel_name(kernel, a_col, b_col, dtype), [0, 0.0, 0.0, 0])
            a[0] += 1
            a[1] += flops.value
            a[2] += ms.value
            a[3] += splitk > 1
        return {k: {'launches': v[0], 'avg_us': 1e3 * v[2] / v[0], 'tflops': v[1] / (v[2] * 1e-3) / 1e12,
                    'total_ms': v[2], 'flops_per_launch': v[1] / v[0], 'splitk_launches': v[3]}
                for k, v in agg.items()}


def committed_pmc_traffic(kernel_key: str):
    """HBM bytes per launch of `kernel_key` from the newest committed PMC passes (profiles/*_hbm_traffic_pmc.json), or
    None: the fallback when the live measurement below is switched off or fails"""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, 'profiles', '*_hbm_traffic_pmc.json')))
    if not files:
        return None
    k = json.load(open(files[-1]))['kernels'].get(kernel_key)
    return k['hbm_bytes_per_launch_corrected'] if k else None


def live_pmc_traffic(kernel_key: str, workload: str, dtype: str, timeout_s: float = 150.0):
    """HBM bytes per launch of `kernel_key`, measured NOW on this box: two child runs of this same script (3 steps of the
    same workload) under `rocprofv3 --kernel-trace --pmc FETCH_SIZE` and `--pmc WRITE_SIZE` — separate passes, counters
    in KiB, FETCH_SIZE doubled (gfx950 tallies its 128-byte requests at 64 bytes), exactly as MI355X_MICROARCH.md's
    HBM / rocprofv3 section prescribes.  The counters sit at the L2 <-> fabric boundary (Infinity-Cache hits included).
    Hardware counters cannot be read from inside the measured process, hence the children; the parent keeps running
    (nothing is exec'ed).  -> (bytes or None, how)"""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile
    if shutil.which('rocprofv3') is None:
        return None, 'rocprofv3 not found'
    per = {}
    for counter in ('FETCH_SIZE', 'WRITE_SIZE'):
        d = tempfile.mkdtemp(prefix='pk_pmc_', dir='/tmp')
        cmd = ['rocprofv3', '--kernel-trace', '--pmc', counter, '--output-format', 'csv', '-d', d, '--', sys.executable,
               os.path.abspath(__file__), '--workload', workload, '--dtype', dtype, '--steps', '3', '--warmup', '1',
               '--no-cpu-baseline', '--no-roofline', '--no-live-traffic', '--no-extra-workloads']
        env = dict(os.environ, TMPDIR='/tmp')
        for k in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE'):
            env.pop(k, None)
        try:
            subprocess.run(cmd, cwd='/tmp', env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL,
                           timeout=timeout_s, check=True)
            n, tot = 0, 0.0
            for f in glob.glob(d + '/**/*counter_collection.csv', recursive=True):
                for r in csv.DictReader(open(f)):
                    name = r['Kernel_Name'].replace('(anonymous namespace)::', '').replace('void ', '').split('(')[0]
                    if r['Counter_Name'] == counter and name == kernel_key:
                        n += 1
                        tot += float(r['Counter_Value'])
            if n == 0:
                return None, f'{counter}: no dispatch of {kernel_key} in the counter pass'
            per[counter] = tot / n
        except Exception as e:  # rocprofv3 missing counters, time-out, ...: report and fall back
            return None, f'{counter} pass failed: {type(e).__name__}'
        finally:
            shutil.rmtree(d, ignore_errors=True)
    return int((2.0 * per['FETCH_SIZE'] + per['WRITE_SIZE']) * 1024), \
        'live: rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE, two separate 3-step child runs of this command on this box'


def cpu_baseline(budget_s: float):
    """the CPU oracle on BASELINE configs[0] (fp32, 8 x (64, 64), ragged=False), fwd+bwd, host cores"""
    import numpy as np
    import paramgen
    from oracle import ref_cpu as O
    from pasero_amd.config import TransformerConfig, DistributedConfig, SyntheticTask
    from pasero_amd.transformer import Transformer
    cfg = TransformerConfig(dropout=0.0)
    V, B, S, T = 8032, 8, 64, 64
    model = Transformer(cfg, DistributedConfig(), SyntheticTask(V))  # only to enumerate names / shapes (CPU, no compute)
    names_shapes = [(k, tuple(v.shape)) for k, v in model.state_dict().items()]
    del model
    P = O.to_torch_state(paramgen.make_state_dict(1, names_shapes))
    P['decoder.embed_tokens.weight'] = P['encoder.embed_tokens.weight']
    for v in P.values():
        v.requires_grad_()
    b = paramgen.make_text_batch(1, B, S, T, V, ragged=False)
    tb = {k: torch.from_numpy(v) for k, v in b.items()}

    def step():
        for v in P.values():
            v.grad = None
        loss, logs = O.transformer_forward(P, cfg, **tb)
        loss.backward()
        return logs['num_tokens']

    step()  # warm-up (first call pages in the kernels)
    # a batch of 512 tokens cannot feed every core of a many-socket host: use the thread count that is fastest
    best = (None, 1e9)
    tried, trial = [], {}
    for nthr in sorted({8, 16, 32, 64, torch.get_num_threads()}):
        if nthr > (os.cpu_count() or 1):
            continue
        torch.set_num_threads(nthr)
        t0 = time.perf_counter()
        step()
        dt = time.perf_counter() - t0
        tried.append(nthr)
        trial[nthr] = round(dt, 3)
        if dt < best[1]:
            best = (nthr, dt)
    torch.set_num_threads(best[0])
    t0 = time.perf_counter()
    n, tokens = 0, 0
    while True:
        tokens += step()
        n += 1
        el = time.perf_counter() - t0
        if el >= budget_s or n >= 50:
            break
    return {'value': tokens / el, 'unit': 'target tokens/s', 'cores': torch.get_num_threads(), 'kind': 'port',
            'sample': f'oracle/ref_cpu.py fwd+bwd, Transformer-base fp32, batch 8x(64,64), {n} steps in {el:.1f} s '
                      f'({os.cpu_count()} logical CPUs)',
            'cores_choice': f'{best[0]} threads = the fastest of {tried} on one trial step each (seconds per step: '
                            f'{trial}); a batch of 512 tokens gives torch too little work per thread to gain from more'}


def count_flops(cfg, B: int, S: int, T: int, V: int, trained_encoder_layers=None) -> float:
    """SURVEY §8d algorithmic FLOPs of one fwd+bwd step (2·MACs, bwd = 2x fwd, causal self-attention at half).
    `trained_encoder_layers` (the IWSLT recipe: a frozen backbone): only that many encoder layers have weight gradients —
    forward and the gradient towards the input still run through every layer (the trained frontend sits below them), the
    decoder and the vocabulary projection have none; the adapters' own GEMMs (d x 64) are not counted."""
    d, fe, fd = cfg.embed_dim, cfg.encoder_ffn_dim, cfg.decoder_ffn_dim
    Le, Ld = cfg.encoder_layers, cfg.decoder_layers
    enc_lin, enc_att = B * S * Le * (8 * d * d + 4 * d * fe), B * S * Le * 4 * S * d
    dec_lin = B * T * Ld * (12 * d * d + 4 * d * fd) + B * S * Ld * 4 * d * d
    dec_att = B * T * Ld * (2 * T * d + 4 * S * d)
    voc = B * T * 2 * d * V
    if trained_encoder_layers is None:
        return 3.0 * (enc_lin + enc_att + dec_lin + dec_att + voc)
    return (2.0 * (enc_lin + dec_lin + voc) + enc_lin * trained_encoder_layers / Le + 3.0 * (enc_att + dec_att))


def rehearse_cpu(args, rank: int, world: int):
    """CPU rehearsal of everything around the HIP model in a multi-rank run: rendezvous, the bucketed reducer, the
    fused logs all-reduce, the barrier / max-over-ranks timing protocol and rank 0's JSON line.  Not a measurement."""
    import torch.distributed as dist
    import torch.nn as nn
    from pasero_amd.ddp import DistributedDataParallel, reduce_logs
    if world > 1:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29500')
        dist.init_process_group('gloo', rank=rank, world_size=world)
    torch.manual_seed(1234)
    net = nn.Sequential(nn.Linear(64, 256), nn.ReLU(), nn.Linear(256, 64))
    ddp = DistributedDataParallel(net, bucket_cap_mb=0.02) if world > 1 else net
    x = torch.randn(32, 64, generator=torch.Generator().manual_seed(1 + rank))
    tokens_per_step = 32

    def step():
        for p in net.parameters():
            p.grad = None
        ddp(x).pow(2).sum().backward()
        return reduce_logs({'loss': 1.0, 'nll_loss': 1.0, 'num_tokens': tokens_per_step, 'num_lines': 1})

    def fence():
        if dist.is_initialized():
            dist.barrier()

    for _ in range(args.warmup):
        step()
    fence()
    t0 = time.perf_counter()
    tokens = 0
    for _ in range(args.steps):
        tokens += step()['num_tokens']  # already the sum over the ranks
    fence()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = t.item()
    if rank == 0:
        _JSON_LINE.append(json.dumps({'metric': 'target tokens/sec (fwd+bwd), Transformer-base d=512', 'value': tokens / elapsed,
                          'unit': 'target tokens/s', 'n_gpus': world, 'rccl_ranks': world, 'steps': args.steps,
                          'warmup': args.warmup, 'ms_per_step': 1e3 * elapsed / args.steps, 'higher_is_better': True,
                          'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
                          'rehearsal': 'CPU/gloo stand-in model: plumbing only, NOT a measurement',
                          'config': {'workload': 'rehearsal', 'tokens_per_rank_per_step': tokens_per_step,
                                     'parallelism': f'dp{world}',
                                     'gradient_all_reduce': ddp.describe() if hasattr(ddp, 'describe') else None}}))
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()


_JSON_LINE = []


def main():
    args = parse_args()
    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        raise SystemExit(self_launch(args))  # (before the redirect below: the children inherit the real stdout)
    # rank 0's stdout carries ONE line, the JSON: whatever native libraries print there while the run is set up (RCCL's
    # version banner at communicator creation, for one) goes to stderr instead; the descriptor is restored for the line
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)
    try:
        run(args)
    finally:
        sys.stdout.flush()
        os.dup2(json_fd, 1)
        os.close(json_fd)
        if _JSON_LINE:
            print(_JSON_LINE[0], flush=True)


def run(args):
    rank = int(os.environ.get('RANK', 0))
    local_rank = int(os.environ.get('LOCAL_RANK', 0))
    world = int(os.environ.get('WORLD_SIZE', 1))
    if world != args.gpus:
        raise SystemExit(f'--gpus {args.gpus} but WORLD_SIZE={world}')
    if args.rehearse_cpu:
        return rehearse_cpu(args, rank, world)
    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs an MI355X (no CPU fallback for the product path)')
    ndev = torch.cuda.device_count()
    torch.cuda.set_device(local_rank % ndev)
    device = torch.device('cuda', local_rank % ndev)
    import torch.distributed as dist
    if world > 1 or args.force_ddp:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29500')
        if args.force_ddp:
            os.environ['PASERO_DDP_FORCE_REDUCE'] = '1'
        kw = {'device_id': device} if args.backend == 'nccl' else {}
        dist.init_process_group(args.backend, rank=rank, world_size=world, **kw)

    from pasero_amd import config as C, rng
    from pasero_amd import adapters  # noqa: F401  (registers adapter_transformer: the IWSLT recipe's architecture)
    from pasero_amd.transformer import Transformer  # noqa: F401
    from pasero_amd.ddp import DistributedDataParallel, reduce_logs

    dtype = {'bf16': torch.bfloat16, 'f16': torch.float16, 'f32': torch.float32}[args.dtype]

    def fence():
        if dist.is_initialized():
            dist.barrier()
        torch.cuda.synchronize()

    def measure(workload: str, steps: int, warmup: int, with_ddp: bool):
        """build `workload`, run `warmup` untimed + `steps` timed steps -> (elapsed s, tokens, GemmTimer, cfg, dims, ddp)"""
        cfg_name, V, B, S, T = WORKLOADS[workload]
        cfg, model, batch, wav = build_workload(workload, dtype, device, rank, world)
        ddp = DistributedDataParallel(model) if with_ddp else model

        def step():
            for p in model.parameters():
                p.grad = None
            if wav is not None:
                from pasero_amd import functional as PF
                batch['encoder_input'] = PF.log_mel(wav).to(dtype)
            loss, logs = ddp(**batch)
            loss.backward()
            if dist.is_initialized():  # Trainer.train_step's per-step log exchange (training.py:431), as ONE all-reduce
                logs = reduce_logs(logs)
            return logs['num_tokens']  # N > 1: already the sum over the ranks

        timer = GemmTimer()
        for _ in range(warmup):
            step()
        fence()
        if not args.no_roofline:
            # the sample buffer is sized from a COUNTED step (one more untimed step with every GEMM call sampled): a fixed
            # guess of 400 launches per step filled up at 54-69 % of the timed region of C5 (586) and the IWSLT recipe (759),
            # so their per-kernel averages and `gemm_share_of_step` came from the first part of it only
            timer.start(4096, stride=1)
            step()
            fence()
            timer.stop()
            timer.launches_per_step = timer.samples
            timer.start(timer.launches_per_step * steps // GemmTimer.STRIDE + 64)
        # five sub-windows of the timed region, cut by events on the launch stream (no synchronisation inside the region):
        # one short sample per round cannot show the 3-5 % the boxes differ by, the spread inside a run can
        nwin = 5 if steps >= 10 else 1
        cuts = [round(i * steps / nwin) for i in range(nwin + 1)]
        marks = []
        clock = ClockSampler(device)
        clock.start()
        t0 = time.perf_counter()
        tokens = 0
        for i in range(steps):
            if i in cuts:
                marks.append(torch.cuda.Event(enable_timing=True))
                marks[-1].record()
            tokens += step()
        marks.append(torch.cuda.Event(enable_timing=True))
        marks[-1].record()
        fence()
        elapsed = time.perf_counter() - t0
        timer.sclk = clock.stop()
        w = sorted(marks[k].elapsed_time(marks[k + 1]) / (cuts[k + 1] - cuts[k]) for k in range(nwin))
        timer.windows = {'ms_per_step_min': w[0], 'ms_per_step_median': w[len(w) // 2], 'ms_per_step_max': w[-1],
                         'windows': nwin, 'how': 'HIP events on the launch stream at the window boundaries'}
        if not args.no_roofline:
            timer.stop()
        if world > 1:
            t = torch.tensor([elapsed], dtype=torch.float64, device=device)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            elapsed = t.item()
        return elapsed, tokens, timer, cfg, (cfg_name, V, B, S, T), ddp

    elapsed, tokens, timer, cfg, (cfg_name, V, B, S, T), ddp = measure(args.workload, args.steps, args.warmup,
                                                                       world > 1 or args.force_ddp)

    if rank == 0:
        # SURVEY §8d algorithmic FLOPs of one fwd+bwd batch (speech: the encoder runs on the S/2 subsampled positions;
        # the conv frontend's own FLOPs are not counted)
        step_flops = count_flops(cfg, B, S if args.workload not in SPEECH else S // 2, T, V,
                                 3 if args.workload == 'c4_iwslt' else None)
        out = {
            'metric': 'target tokens/sec (fwd+bwd), Transformer-base d=512',
            'value': tokens / elapsed,  # whole job: the sum over the N GPUs
            'value_per_gpu': tokens / elapsed / world,  # BASELINE's metric is quoted per GPU; identical at N = 1
            'unit': 'target tokens/s',
            'n_gpus': world,
            'rccl_ranks': dist.get_world_size() if dist.is_initialized() else 1,
            'tokens_per_rank_per_step': tokens / world / args.steps,
            'steps': args.steps,
            'warmup': args.warmup,
            'ms_per_step': 1e3 * elapsed / args.steps,
            'ms_per_step_windows': timer.windows,   # spread inside the timed region (rank 0)
            'sclk_mhz': timer.sclk,                 # the shader clock held during it (rank 0)
            'higher_is_better': True,
            'scaling': 'weak',
            'vs_baseline': None,
            'dtype': args.dtype,
            'data': 'synthetic',
            'config': {'workload': f'{args.workload}: {cfg_name} V={V}, per-GPU batch (B,S,T)=({B},{S},{T}), '
                                   f'dropout {cfg.dropout}, label smoothing {cfg.label_smoothing}, full-length rows',
                       'global_batch': B * world, 'seq_len': T, 'parallelism': f'dp{world}',
                       # the reducer describing itself (transport, the schedule chosen and the measured time of each of the
                       # three, every bucket's size in launch order): a scaling run must explain its own numbers
                       'gradient_all_reduce': (ddp.describe() if hasattr(ddp, 'describe') else None),
                       'step': 'forward + backward' + (' + bucketed RCCL all-reduce + fused logs all-reduce' if world > 1 else ''),
                       'algorithmic_tflop_per_step_per_gpu': step_flops / 1e12,
                       'model_tflops_per_gpu': step_flops * args.steps / elapsed / 1e12,
                       'mfma_peak_fraction_whole_step': step_flops * args.steps / elapsed / 1e12 / PEAK_BF16_TFLOPS},
        }
        out['config']['sclk_mhz_median'] = timer.sclk['median'] if timer.sclk else None
        roofline = None
        if not args.no_roofline and timer.samples:
            summ = timer.summary()
            dom = max(summ, key=lambda k: summ[k]['total_ms'])
            d = summ[dom]
            roofline = {'bound': 'mfma', 'kernel': dom, 'achieved': d['tflops'], 'peak': PEAK_BF16_TFLOPS,
                               'unit': 'TFLOP/s', 'frac': d['tflops'] / PEAK_BF16_TFLOPS, 'traffic': None,
                               'avg_launch_us': d['avg_us'], 'sampled_launches': d['launches'],
                               'sampling': f'HIP events around the kernel of every {GemmTimer.STRIDE}th pk_gemm call, on its stream',
                               'flops_per_launch': d['flops_per_launch'],
                               'gemm_launches_per_step': timer.launches_per_step,
                               'gemm_share_of_step': GemmTimer.STRIDE * sum(v['total_ms'] for v in summ.values()) / (1e3 * elapsed),
                               'all_gemm_kernels': {k: {'tflops': round(v['tflops'], 1), 'avg_us': round(v['avg_us'], 1),
                                                        'sampled_launches': v['launches']} for k, v in summ.items()}}
            traffic, how = (None, 'live measurement off') if (args.no_live_traffic or world > 1) else \
                live_pmc_traffic(dom, args.workload, args.dtype)
            if traffic is None:
                traffic, how = committed_pmc_traffic(dom), f'committed profiles/ ({how})'
            roofline['traffic'] = traffic
            roofline['traffic_source'] = how
            if traffic:  # the same kernel against the memory roof: L2 <-> fabric bytes per launch / its duration
                gbps = traffic / (d['avg_us'] * 1e-6) / 1e9
                roofline['hbm'] = {'achieved': gbps, 'peak': PEAK_HBM_GBPS, 'unit': 'GB/s', 'frac': gbps / PEAK_HBM_GBPS}
        if world == 1 and not args.no_extra_workloads and args.workload == 'c2_base_bf16':
            # the other BASELINE configurations that fit one GPU, timed by THIS run after the headline region (same
            # process, same protocol, a few steps each): d = 1024 is where north_star states its 40 % target
            del ddp
            out['extra_workloads'] = {}
            # the same numbers, compact, INSIDE `config` (the driver keeps `config` whole but only a tail of the line: C3's
            # record was cut out of BENCH_r04 / r05): name -> [ms per step, whole-step fraction of 2.5 PFLOP/s, sclk MHz]
            out['config']['extras'] = {}
            for name in ('c3_big', 'c4_whisper', 'c5_nllb_1b3', 'c4_iwslt'):
                torch.cuda.empty_cache()
                e_steps = args.extra_steps if name not in DEVICE_INIT else max(1, args.extra_steps // 2)  # (80-100 ms steps)
                e_el, e_tok, e_timer, e_cfg, (_, eV, eB, eS, eT), _m = measure(name, e_steps, 3, False)
                del _m
                fl = count_flops(e_cfg, eB, eS if name not in SPEECH else eS // 2, eT, eV, 3 if name == 'c4_iwslt' else None)
                rec = {'config': f'{WORKLOADS[name][0]} V={eV}, batch (B,S,T)=({eB},{eS},{eT}), {args.dtype}',
                       'steps': e_steps, 'warmup': 3, 'ms_per_step': 1e3 * e_el / e_steps,
                       'value': e_tok / e_el, 'unit': 'target tokens/s',
                       'algorithmic_tflop_per_step': fl / 1e12,
                       'mfma_peak_fraction_whole_step': fl * e_steps / e_el / 1e12 / PEAK_BF16_TFLOPS,
                       'ms_per_step_windows': e_timer.windows, 'sclk_mhz': e_timer.sclk}
                if name == 'c4_iwslt':
                    rec['note'] = ('examples/IWSLT2023 recipe: frozen NLLB-1.3B backbone, trainable in_linear + conv k5 + encoder '
                                   'layers 0-2 + adapters; FLOPs count forward + input gradients everywhere, weight gradients '
                                   'only where parameters train')
                if not args.no_roofline and e_timer.samples:
                    es = e_timer.summary()
                    edom = max(es, key=lambda k: es[k]['total_ms'])
                    top = sorted(es, key=lambda k: -es[k]['total_ms'])[:4]  # (the full tables: profiles/rNN_<workload>_*)
                    rec.update({'dominant_kernel': edom, 'achieved_tflops': es[edom]['tflops'],
                                'frac': es[edom]['tflops'] / PEAK_BF16_TFLOPS, 'avg_launch_us': es[edom]['avg_us'],
                                'sampled_launches': es[edom]['launches'],
                                'gemm_launches_per_step': e_timer.launches_per_step,
                                'gemm_share_of_step': GemmTimer.STRIDE * sum(v['total_ms'] for v in es.values()) / (1e3 * e_el),
                                'top_gemm_kernels': {k: {'tflops': round(es[k]['tflops'], 1), 'avg_us': round(es[k]['avg_us'], 1),
                                                         'sampled_launches': es[k]['launches']} for k in top}})
                out['extra_workloads'][name] = rec
                frac = round(rec['mfma_peak_fraction_whole_step'], 4)
                sclk = round(e_timer.sclk['median']) if e_timer.sclk else None
                out['config']['extras'][name] = [round(rec['ms_per_step'], 3), frac, sclk]
                out['config'][name + '_ms_per_step'] = round(rec['ms_per_step'], 3)  # (flat copies: scalars survive any parser)
                out['config'][name + '_mfma_frac_whole_step'] = frac
        if roofline is not None:  # (after the extras: the line's longest table last but one)
            out['roofline'] = roofline
        if world == 1 and not args.no_cpu_baseline:
            out['cpu_baseline'] = cpu_baseline(args.cpu_seconds)
        _JSON_LINE.append(json.dumps(out))
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
