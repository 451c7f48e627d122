"""CPU tests of the drop-in boundary: the C-ABI library loads and exports every symbol of include/pasero_hip.h, the
host-side mirror exposes the reference's parameter names / shapes (= checkpoint keys) and class surface, and the
product path fails loudly without a GPU (no fallback)."""
import os
import re

import pytest
import torch

from conftest import ROOT, load_golden, golden_names_shapes
from model_utils import build_model


def test_library_exports_every_header_symbol():
    from pasero_amd import lib
    L = lib.load()
    header = open(os.path.join(ROOT, 'include', 'pasero_hip.h')).read()
    names = set(re.findall(r'\b(pk_[a-z0-9_]+)\s*\(', header))
    assert len(names) >= 20
    for n in sorted(names):
        assert hasattr(L, n), f'{n} declared in include/pasero_hip.h but not exported'
    assert names == set(lib.SIGNATURES), 'pasero_amd/lib.py SIGNATURES out of sync with the header'
    # ... and nothing else: the launchers the translation units call across files stay local (linker version script
    # generated from the header, csrc/Makefile: exports.map)
    import subprocess
    out = subprocess.run(['nm', '-D', '--defined-only', lib.LIB_PATH], capture_output=True, text=True, check=True).stdout
    exported = {ln.split()[-1] for ln in out.splitlines() if ' T ' in ln}
    assert exported == names, f'exported but not declared: {sorted(exported - names)}'
    assert L.pk_version() >= 100


@pytest.mark.parametrize('name', ['tiny_encdec_post', 'tiny_encdec_pre', 'tiny_encdec_rotary', 'tiny_encdec_swiglu', 'speech_whisper',
                                  'speech_iwslt', 'base_c1', 'tiny_adapter', 'tiny_lora', 'tiny_hd128', 'tiny_encdec_rms', 'tiny_opts_a', 'tiny_opts_b', 'tiny_hd128_rotary', 'tiny_lora_rotary', 'tiny_freeze_embed', 'tiny_freeze_shared'])
def test_parameter_names_and_shapes_match_reference(name):
    g = load_golden(name)
    _, model = build_model(g)
    ours = [(k, tuple(v.shape)) for k, v in model.state_dict().items()]
    assert ours == golden_names_shapes(g)


def test_base_model_parameter_count():
    g = load_golden('base_c1')
    _, model = build_model(g)
    assert model.total_param_count == 48_250_880  # SURVEY §8c / examples/TED-top20/training-en-centric.yaml:6
    assert len(list(model.state_dict())) == len(golden_names_shapes(g))


def test_no_cpu_fallback():
    g = load_golden('tiny_encdec_post')
    _, model = build_model(g)
    from model_utils import text_batch
    with pytest.raises(RuntimeError, match='no CPU fallback'):
        model(**text_batch(g))


def test_class_surface():
    from pasero_amd import transformer as T, modules as M
    for n in ('Transformer', 'TransformerEncoder', 'TransformerDecoder', 'TransformerEncoderLayer',
              'TransformerDecoderLayer', 'Encoder', 'Decoder', 'EncoderDecoder', 'DummyEncoder', 'BaseModel'):
        assert hasattr(T, n)
    for n in ('MultiheadAttention', 'Linear', 'Embedding', 'SinusoidalPositionalEmbedding',
              'LearnedPositionalEmbedding', 'ConvolutionSubsampler', 'fast_init', 'set_tp_group',
              'set_sequence_parallel', 'checkpoint_wrapper', 'get_activation_fn', 'Identity', 'WrappableLinear'):
        assert hasattr(M, n)
    for hook in ('ffn', 'self_attention', 'self_attn_residual', 'self_attn_prenorm', 'self_attn_postnorm',
                 'ffn_residual', 'ffn_prenorm', 'ffn_postnorm'):
        assert hasattr(T.TransformerEncoderLayer, hook) and hasattr(T.TransformerDecoderLayer, hook)
    for hook in ('cross_attention', 'cross_attn_residual', 'cross_attn_prenorm', 'cross_attn_postnorm'):
        assert hasattr(T.TransformerDecoderLayer, hook)
    with pytest.raises(NotImplementedError):
        M.set_tp_group(object())


def test_benchmark_names_of_the_reference_are_kept():
    """`pasero-train --benchmark` (cli/train.py:109-110) logs <name>_wall / _mem for the blocks the reference wraps in
    `utils.benchmark(name)`: 'attention' (modules.py:578), 'loss' (transformer.py:323), 'encoder' / 'decoder'
    (transformer.py:697,830), 'output_projection' (transformer.py:892).  The mirror classes carry the same names on the
    same methods; stand-alone, the object has the reference's surface and metric names."""
    import inspect
    from pasero_amd import modules, profiling, transformer
    wrapped = {'attention': modules.MultiheadAttention.forward, 'encoder': transformer.TransformerEncoder.forward,
               'decoder': transformer.TransformerDecoder.forward, 'loss': transformer.Transformer.compute_loss}
    for name, fn in wrapped.items():
        assert hasattr(fn, '__wrapped__'), name
        src = inspect.getsource(fn)
        assert f"_bench_region('{name}')" in src, name
    assert "_bench_block('output_projection')" in inspect.getsource(transformer.TransformerDecoder.forward)
    b = profiling.Benchmark(use_cuda=False, enabled=True)
    with b('forward'):
        with b('forward'):  # a nested block of the same name counts once
            pass
    with b.pause():
        with b('skipped'):
            pass
    m = b.metrics
    assert set(m) == {'forward_wall'} and m['forward_wall'] >= 0
    for attr in ('enable', 'disable', 'pause', 'reset', 'cpu', 'metrics'):
        assert hasattr(profiling.benchmark, attr)
    calls = []

    @profiling.region('x')
    def f(a):
        calls.append(a)
        return a + 1
    assert f(1) == 2 and calls == [1]  # disabled: a plain call


def test_gemm8p_assembly_audit_is_part_of_the_build():
    """csrc/gemm8p.hip keeps an LDS-DMA prefetch in flight behind hand-counted `s_waitcnt vmcnt(6)` and reads col-form
    operands with inline-asm `ds_read_b64_tr_b16`: the build audits its assembly (tools/check_asm_loads.py) and this
    test checks the audit ran on the current source and found every instantiation clean"""
    import os
    import subprocess
    from conftest import ROOT
    csrc = os.path.join(ROOT, 'pasero_amd', 'csrc')
    # gemm8p.hip: 3 operand layouts x 2 dtypes x {lean, general epilogue} x {whole, partial last K-tile} + the grouped
    # weight-gradient kernel per dtype; gemmln.hip (the same K-loop discipline on a 128 x 512 tile): one kernel per dtype
    # attention_long.hip (tools/check_asm_dma.py): the tile prefetch of its three kernels is not waited for where it is requested
    for name, at_least in (('gemm8p', 26), ('gemmln', 2), ('attention_long', 12)):
        subprocess.check_call(['make', '-C', csrc, f'{name}.audit'], stdout=subprocess.DEVNULL)
        report = open(os.path.join(csrc, f'{name}.audit')).read()
        last = report.strip().splitlines()[-1]
        assert last.endswith(', 0 problems') and int(last.split()[0]) >= at_least, (name, last)
        assert 'PROBLEM' not in report


def test_assembly_audit_tells_loop_spills_from_spills_that_run_once(tmp_path):
    """tools/check_asm_loads.py rule 3: scratch traffic in a block LLVM marks as part of a loop is refused; a spill between
    the loop markers but outside every loop block (ahead of the first iteration / behind the last) is only reported"""
    import os
    import subprocess
    import sys
    from conftest import ROOT
    tool = os.path.join(ROOT, 'tools', 'check_asm_loads.py')
    head = ['_ZN12_GLOBAL__N_113gemm8p_kernelIfEEvv:', '\t; PK8P_LOOP_BEGIN']
    tail = ['\t; PK8P_LOOP_END', '\ts_endpgm', '.Lfunc_end0:']
    loop = ['.LBB0_1:                                ; =>This Inner Loop Header: Depth=1', '\ts_waitcnt vmcnt(6)',
            '\tv_mfma_f32_32x32x16_bf16 v[0:15], v[16:19], v[20:23], v[0:15]', '\ts_cbranch_scc1 .LBB0_1']
    spill = '\tscratch_store_dword off, v230, off      ; 4-byte Folded Spill'

    def run(lines):
        f = tmp_path / 'k.s'
        f.write_text('\n'.join(lines) + '\n')
        return subprocess.run([sys.executable, tool, str(f)], capture_output=True, text=True)

    once = run(head + [spill] + loop + ['.LBB0_2:', '\tscratch_load_dword v230, off, off'] + tail)
    assert once.returncode == 0 and '0 problems' in once.stdout and 'runs once' in once.stdout, once.stdout
    inside = run(head + loop[:2] + [spill] + loop[2:] + tail)
    assert inside.returncode == 1 and 'spill traffic inside the K loop' in inside.stdout, inside.stdout
    # an assembly without LLVM's loop annotations cannot be audited for in-loop spills: refused, not waved through
    bare = [ln.split(';')[0].rstrip() if ln.startswith('.LBB') else ln for ln in loop]
    unannotated = run(head + bare[:2] + [spill] + bare[2:] + tail)
    assert unannotated.returncode == 1 and 'no label with a loop annotation' in unannotated.stdout, unannotated.stdout


def test_ctypes_structures_have_the_layout_of_the_header(tmp_path):
    """the structs that cross the C ABI by pointer (PkLayer and its blocks, PkWgradProblem, the decoder plan) are declared twice —
    include/pasero_hip.h and the ctypes mirrors in lib.py / decode.py: compiled from the header by gcc, every field must sit at
    the offset ctypes gives it (a field added on one side only shifts everything behind it silently)"""
    import ctypes
    import subprocess
    from pasero_amd import lib, decode
    mirrors = {'PkWgradProblem': lib.PkWgradProblem, 'PkAttnBlock': lib.PkAttnBlock, 'PkFfnBlock': lib.PkFfnBlock,
               'PkLayer': lib.PkLayer, 'PkDecoderLayerWeights': decode.PkDecoderLayerWeights, 'PkDecoderPlan': decode.PkDecoderPlan}
    rename = {'self_': 'self'}  # (`self` is spelled `self_` on the Python side)
    lines = []
    for name, cls in mirrors.items():
        lines.append(f'printf("{name} %zu\\n", sizeof({name}));')
        for field, _ in cls._fields_:
            c = rename.get(field, field)
            lines.append(f'printf("{name}.{field} %zu\\n", offsetof({name}, {c}));')
    src = tmp_path / 'layout.c'
    src.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "pasero_hip.h"\nint main(void) {\n' + '\n'.join(lines) + '\nreturn 0; }\n')
    exe = tmp_path / 'layout'
    subprocess.run(['gcc', '-I', os.path.join(ROOT, 'include'), str(src), '-o', str(exe)], check=True)
    got = dict(line.split() for line in subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.splitlines())
    for name, cls in mirrors.items():
        assert int(got[name]) == ctypes.sizeof(cls), (name, got[name], ctypes.sizeof(cls))
        for field, _ in cls._fields_:
            assert int(got[f'{name}.{field}']) == getattr(cls, field).offset, (name, field)
