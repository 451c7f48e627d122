#!/usr/bin/env python3
"""Build-container audit of the drop-in boundary against the REAL reference tree (/root/reference, read-only):
  1. every `cfg.<attr>` / `dist_cfg.<attr>` pasero_amd reads exists on the reference's configuration classes;
  2. every preset in pasero_amd/config.py has the reference's defaults (pasero/config.py MODEL_CONFIGS);
  3. with the reference's own configuration objects and registry, `get_architecture` resolves to the pasero_amd classes
     and their state_dict names/shapes equal the reference model's.
No forward pass runs (there is no GPU here and pasero_amd has no CPU path).  Exit code 0 = clean.
Test infrastructure only; the reference never travels to the GPU box, so this runs only where /root/reference exists.
"""
import dataclasses
import glob
import os
import re
import sys
import types

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = '/root/reference'
sys.dont_write_bytecode = True


def import_reference():
    # the two third-party imports absent from the image, stubbed as in oracle/make_golden.py
    sb, sbm = types.ModuleType('sacrebleu'), types.ModuleType('sacrebleu.metrics')
    sbm.METRICS = {'BLEU': type('B', (), {'TOKENIZERS': {}})}
    sb.metrics = sbm
    sys.modules['sacrebleu'], sys.modules['sacrebleu.metrics'] = sb, sbm
    names = ['stopes', 'stopes.pipelines', 'stopes.pipelines.monolingual', 'stopes.pipelines.monolingual.utils',
             'stopes.pipelines.monolingual.utils.text_normalizer']
    for n in names:
        sys.modules[n] = types.ModuleType(n)
    sys.modules[names[-1]].remove_non_printing_char = lambda s: s
    sys.modules[names[-1]].replace_unicode_punct = lambda s: s
    sys.path.insert(0, REF)
    sys.path.insert(0, REPO)
    from pasero import config as RC
    from pasero.models import transformer as RT, adapters as RA
    return RC, RT, RA


class FakeTask:
    freeze_encoder_embed_mask = None

    def __init__(self, enc, dec):
        self.encoder_num_embeddings, self.decoder_num_embeddings = enc, dec


def state_dict_plumbing(RC, RT):
    """remap_state_dict / update_state_dict / clean_state_dict (transformer.py:382-497,584-592) applied by the reference
    model and by ours to the same synthetic checkpoints must leave the same keys and values, and ours must then load
    strictly"""
    import copy
    import torch
    import pasero_amd.transformer as T
    problems = []

    class Task(FakeTask):
        def remap_encoder_embed(self, e): return e * 2
        def remap_decoder_embed(self, e): return e * 3

    def small(**kw):
        cfg = RC.TransformerConfig(embed_dim=128, encoder_ffn_dim=64, decoder_ffn_dim=64, encoder_attention_heads=2,
                                   decoder_attention_heads=2, encoder_layers=2, decoder_layers=2, **kw)
        cfg.label_smoothing, cfg.model_type, cfg.decoder_max_len = 0.1, 'encoder_decoder', 64
        return cfg

    def stale_checkpoint(model):
        """what old fairseq / HuggingFace checkpoints look like"""
        sd = {k: v.clone() for k, v in model.state_dict().items()}
        for layer in ('encoder.layers.0.self_attn', 'decoder.layers.1.encoder_attn'):
            for kind in ('weight', 'bias'):
                parts = [sd.pop(f'{layer}.{p}_proj.{kind}') for p in 'qkv']
                sd[f'{layer}.in_proj_{kind}'] = torch.cat(parts)
        sd['decoder.version'] = torch.tensor([3.0])
        sd['lm_head.weight'] = torch.zeros(3, 3)
        sd.pop('decoder.embed_tokens.weight', None)
        if 'decoder.layer_norm.weight' in sd:
            for kind in ('weight', 'bias'):
                sd[f'decoder.final_layer_norm.{kind}'] = sd.pop(f'decoder.layer_norm.{kind}')
        if model.cfg.tied_output_projection:
            sd['decoder.output_projection.weight'] = torch.zeros(5, 128)
        sd['encoder.embed_tokens.frozen_embedding.weight'] = torch.zeros(2, 2)
        return sd

    cases = [('post-norm, training', small(), True), ('pre-norm, inference', small(encoder_prenorm=True,
             decoder_prenorm=True), False), ('untied, unshared', small(tied_output_projection=False,
             shared_embeddings=False), True), ('lora training', small(lora_rank=4), True),
             ('lora inference', small(lora_rank=4), False), ('shifted layers', small(), True)]
    for tag, cfg, training in cases:
        if tag == 'shifted layers':
            cfg.shift_encoder_layers, cfg.shift_decoder_layers = 1, 0
        torch.manual_seed(0)
        ref = RT.Transformer(cfg, RC.DistributedConfig(), Task(50, 50)).train(training)
        ours = T.Transformer(cfg, RC.DistributedConfig(), Task(50, 50)).train(training)
        ours.load_state_dict(ref.state_dict())
        base = stale_checkpoint(ref)
        if tag.startswith('lora') and training:  # a checkpoint of the backbone only: LoRA parameters are new
            base = {k: v for k, v in base.items() if '.lora.' not in k}
        a, b = copy.deepcopy(base), copy.deepcopy(base)
        for model, sd in ((ref, a), (ours, b)):
            model.remap_state_dict(sd)
            model.update_state_dict(sd)
        if set(a) != set(b):
            problems.append(f'state dict plumbing [{tag}]: keys differ: only reference {sorted(set(a) - set(b))[:5]}, '
                            f'only ours {sorted(set(b) - set(a))[:5]}')
            continue
        lora_new = tag == 'lora training'
        bad = [k for k in a if a[k].shape != b[k].shape or
               (not (lora_new and '.lora.' in k) and not torch.allclose(a[k].float(), b[k].float(), atol=1e-6))]
        if bad:
            problems.append(f'state dict plumbing [{tag}]: values differ for {bad[:5]}')
        if tag != 'shifted layers':
            ref.load_state_dict(a, strict=True)
            try:
                ours.load_state_dict(b, strict=True)
            except Exception as e:  # noqa: BLE001
                problems.append(f'state dict plumbing [{tag}]: strict load fails: {str(e)[:300]}')
        a, b = dict(ref.state_dict()), dict(ours.state_dict())
        ref.clean_state_dict(a)
        ours.clean_state_dict(b)
        if set(a) != set(b):
            problems.append(f'clean_state_dict [{tag}]: keys differ')
    problems += adapter_plumbing(RC, Task)
    return problems


def adapter_plumbing(RC, Task):
    """AdapterTransformer.update_state_dict / clean_state_dict (adapters.py:144-166): new adapters keep their
    initialisation when training on a backbone checkpoint, adapters the model does not hold are parked in
    `extra_adapters` and written back by clean_state_dict, missing adapters are disabled at inference"""
    import copy
    import torch
    from pasero.models import adapters as RA
    import pasero_amd.adapters as A
    problems = []

    def cfg_for(enc, dec):
        cfg = RC.AdapterTransformerConfig(embed_dim=128, encoder_ffn_dim=64, decoder_ffn_dim=64,
                                          encoder_attention_heads=2, decoder_attention_heads=2, encoder_layers=2,
                                          decoder_layers=1, encoder_adapters=enc, decoder_adapters=dec)
        cfg.label_smoothing, cfg.model_type, cfg.decoder_max_len = 0.1, 'encoder_decoder', 64
        return cfg

    torch.manual_seed(1)
    donor = RA.AdapterTransformer(cfg_for(['a', 'x'], ['a', 'x']), RC.DistributedConfig(), Task(40, 40))
    full = {k: v.clone() for k, v in donor.state_dict().items()}
    backbone = {k: v for k, v in full.items() if '.adapters.' not in k}
    for tag, enc, dec, training, ckpt in (('new adapters on a backbone', ['a'], ['a', 'b'], True, backbone),
                                          ('checkpoint holds other adapters', ['a'], ['a'], True, full),
                                          ('inference with a subset', ['a'], ['x', 'zz'], False, full)):
        cfg = cfg_for(enc, dec)
        ref = RA.AdapterTransformer(cfg, RC.DistributedConfig(), Task(40, 40)).train(training)
        ours = A.AdapterTransformer(cfg, RC.DistributedConfig(), Task(40, 40)).train(training)
        ours.load_state_dict(ref.state_dict())
        a, b = copy.deepcopy(ckpt), copy.deepcopy(ckpt)
        ref.update_state_dict(a)
        ours.update_state_dict(b)
        if set(a) != set(b):
            problems.append(f'adapter plumbing [{tag}]: keys differ: only reference {sorted(set(a) - set(b))[:4]}, '
                            f'only ours {sorted(set(b) - set(a))[:4]}')
            continue
        bad = [k for k in a if not torch.equal(a[k], b[k])]
        if bad:
            problems.append(f'adapter plumbing [{tag}]: values differ for {bad[:4]}')
        if sorted(ref.extra_adapters) != sorted(ours.extra_adapters):
            problems.append(f'adapter plumbing [{tag}]: extra_adapters differ')
        ref.load_state_dict(a, strict=True)
        try:
            ours.load_state_dict(b, strict=True)
        except Exception as e:  # noqa: BLE001
            problems.append(f'adapter plumbing [{tag}]: strict load fails: {str(e)[:300]}')
            continue
        a, b = dict(ref.state_dict()), dict(ours.state_dict())
        ref.clean_state_dict(a)
        ours.clean_state_dict(b)
        if set(a) != set(b):
            problems.append(f'adapter clean_state_dict [{tag}]: keys differ: {sorted(set(a) ^ set(b))[:6]}')
        frozen = lambda m: sorted(k for k, p in m.named_parameters() if not p.requires_grad)  # noqa: E731
        if frozen(ref) != frozen(ours):
            problems.append(f'adapter plumbing [{tag}]: frozen parameter sets differ after loading')
    return problems


def main() -> int:
    RC, RT, RA = import_reference()
    import pasero_amd.config as MC
    import pasero_amd.transformer   # noqa: F401  registers 'transformer' in the reference's registry
    import pasero_amd.adapters      # noqa: F401  registers 'adapter_transformer' (after the reference's own import)
    problems = []

    used, dused = set(), set()
    for f in glob.glob(os.path.join(REPO, 'pasero_amd', '*.py')):
        src = open(f).read()
        dused |= set(re.findall(r'\bdist_cfg\.(\w+)', src))
        if not f.endswith('config.py'):
            used |= set(re.findall(r'\bcfg\.(\w+)', src))
    ref_cfg, ref_dist = RC.AdapterTransformerConfig(), RC.DistributedConfig()
    problems += [f'cfg.{a} is read but the reference configuration has no such field' for a in sorted(used)
                 if not hasattr(ref_cfg, a)]
    problems += [f'dist_cfg.{a} is read but the reference has no such field' for a in sorted(dused)
                 if not hasattr(ref_dist, a)]

    task_defaults = ('label_smoothing', 'model_type', 'decoder_max_len')  # set per task (config.py:1146-1153,1241-1272)
    for name, mine_cls in MC.CONFIGS.items():
        ref_cls = RC.MODEL_CONFIGS.get(name)
        if ref_cls is None:
            problems.append(f'preset {name} does not exist in the reference')
            continue
        r, m = ref_cls(), mine_cls()
        for f in dataclasses.fields(m):
            if f.name not in task_defaults and getattr(r, f.name, '<missing>') != getattr(m, f.name):
                problems.append(f'preset {name}.{f.name}: {getattr(m, f.name)!r} here, '
                                f'{getattr(r, f.name, "<missing>")!r} in the reference')

    def sig(model):
        return [(k, tuple(v.shape)) for k, v in model.state_dict().items()]

    cases = [
        (RC.TransformerSmallConfig, dict(shared_embeddings=False), FakeTask(300, 400), RT.Transformer),
        (RC.MBARTConfig, dict(encoder_layers=2, decoder_layers=2), FakeTask(1000, 1000), RT.Transformer),
        (RC.WhisperConfig, dict(encoder_layers=2, decoder_layers=2), FakeTask(0, 51865), RT.Transformer),
        (RC.AdapterTransformerConfig, dict(encoder_adapters=['a'], decoder_adapters=['a', 'b']), FakeTask(500, 500),
         RA.AdapterTransformer),
        (RC.TransformerConfig, dict(lora_rank=4, encoder_layers=1, decoder_layers=1), FakeTask(200, 200), RT.Transformer),
    ]
    for cfg_cls, kw, task, ref_model_cls in cases:
        cfg = cfg_cls(**kw)
        cfg.label_smoothing = 0.1 if cfg.label_smoothing is None else cfg.label_smoothing
        cfg.model_type = cfg.model_type or 'encoder_decoder'
        cfg.decoder_max_len = cfg.decoder_max_len or 256
        arch = RC.get_architecture(cfg)
        if not arch.__module__.startswith('pasero_amd'):
            problems.append(f'{cfg_cls.__name__}: the reference registry resolves to {arch.__module__}.{arch.__name__}')
            continue
        ours = arch(cfg, RC.DistributedConfig(), task)
        theirs = ref_model_cls(cfg, RC.DistributedConfig(), task)
        if sig(ours) != sig(theirs):
            problems.append(f'{cfg_cls.__name__}: state_dict names/shapes differ from {ref_model_cls.__name__}')
        frozen = lambda m: sorted(k for k, p in m.named_parameters() if not p.requires_grad)  # noqa: E731
        if frozen(ours) != frozen(theirs):
            problems.append(f'{cfg_cls.__name__}: the set of frozen parameters differs')
    problems += state_dict_plumbing(RC, RT)
    for p in problems:
        print('AUDIT:', p)
    print(f'audit_against_reference: {len(problems)} problem(s), {len(used)} cfg fields, {len(MC.CONFIGS)} presets, '
          f'{len(cases)} model layouts')
    return 1 if problems else 0


if __name__ == '__main__':
    sys.exit(main())
