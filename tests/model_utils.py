"""helpers shared by the model-level tests"""
import json

import numpy as np
import torch

import paramgen
from conftest import load_golden, golden_names_shapes


def build_cfg(g):
    from pasero_amd.config import TransformerConfig, AdapterTransformerConfig
    arch = str(g['arch']) if 'arch' in getattr(g, 'files', g) else 'transformer'
    cls = AdapterTransformerConfig if arch == 'adapter_transformer' else TransformerConfig
    return cls(**json.loads(str(g['cfg'])))


def build_model(g, dtype=torch.float32, device='cpu'):
    """pasero_amd Transformer with the fixture's config and the deterministic paramgen weights"""
    from pasero_amd.config import DistributedConfig, SyntheticTask, get_architecture
    from pasero_amd import transformer, adapters  # noqa: F401  (registers the architectures)
    cfg = build_cfg(g)
    V = int(g['V'])
    task = SyntheticTask(V)
    if 'freeze_seed' in getattr(g, 'files', g):  # partially frozen source embeddings (tasks/translation.py:141-146)
        task.freeze_encoder_embed_mask = torch.from_numpy(paramgen.make_freeze_mask(int(g['freeze_seed']), V))
    model = get_architecture(cfg)(cfg, DistributedConfig(), task)
    load_paramgen(model, int(g['seed']))
    return cfg, model.to(dtype).to(device)


def load_paramgen(model, seed):
    sd = model.state_dict()
    names_shapes = [(k, tuple(v.shape)) for k, v in sd.items()]
    new = paramgen.make_state_dict(seed, names_shapes)
    first = {}
    for k, v in sd.items():  # tied tensors: value of the FIRST alias (same rule as oracle/make_golden.py)
        new[k] = new[first.setdefault(v.data_ptr(), k)]
    model.load_state_dict({k: torch.from_numpy(v) for k, v in new.items()})


def oracle_state(g, cfg):
    from oracle import ref_cpu as O
    P = O.to_torch_state(paramgen.make_state_dict(int(g['seed']), golden_names_shapes(g)))
    if cfg.shared_embeddings and 'encoder.embed_tokens.weight' in P:
        P['decoder.embed_tokens.weight'] = P['encoder.embed_tokens.weight']
        if 'decoder.embed_tokens.frozen_embedding.weight' in P:
            P['decoder.embed_tokens.frozen_embedding.weight'] = P['encoder.embed_tokens.frozen_embedding.weight']
    if 'freeze_seed' in getattr(g, 'files', g):
        P['encoder.embed_tokens.freeze_mask'] = torch.from_numpy(paramgen.make_freeze_mask(int(g['freeze_seed']), int(g['V'])))
    return P


def text_batch(g, device='cpu'):
    prompt_cols = int(g['prompt_cols']) if 'prompt_cols' in getattr(g, 'files', g) else 0
    b = paramgen.make_text_batch(int(g['seed']), int(g['B']), int(g['S']), int(g['T']), int(g['V']),
                                 prompt_cols=prompt_cols)
    return {k: torch.from_numpy(v).to(device) for k, v in b.items()}


def rel(a, b):
    a = torch.as_tensor(np.asarray(a) if not torch.is_tensor(a) else a).double().cpu()
    b = torch.as_tensor(np.asarray(b) if not torch.is_tensor(b) else b).double().cpu()
    return ((a - b).abs().max() / b.abs().max().clamp_min(1e-30)).item()
