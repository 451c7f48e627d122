"""Random combinations of the configuration switches of the encoder-decoder (norm placement and kind, activation, positions,
head size, biases, tying, prompt loss, label smoothing, layer counts) — the HIP model against the CPU oracle on the same
seeded weights and ragged batches, fp32: loss within 1e-4 (north_star), every gradient within 3e-4.  Each switch has
its own fixture from the real reference (tests/golden); this covers their combinations."""
import numpy as np
import pytest
import torch

import paramgen
from model_utils import load_paramgen, rel
from oracle import ref_cpu as O

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('seed', range(24))
def test_random_configuration_vs_oracle(seed):
    if not torch.cuda.is_available():
        pytest.skip('needs a GPU')
    from pasero_amd.config import TransformerConfig, DistributedConfig, SyntheticTask
    from pasero_amd.transformer import Transformer
    rs = np.random.RandomState(3000 + seed)
    pick = lambda *xs: xs[rs.randint(len(xs))]  # noqa: E731
    hd = pick(64, 64, 128)
    heads = pick(1, 2)
    act = pick('relu', 'gelu', 'gelu_tanh', 'swiglu')
    pos = pick('sinusoidal', 'learned', 'rotary')
    prenorm = bool(rs.randint(2))
    cfg = TransformerConfig(
        embed_dim=hd * heads, encoder_attention_heads=heads, decoder_attention_heads=heads,
        encoder_ffn_dim=pick(128, 192, 320), decoder_ffn_dim=pick(128, 256), encoder_layers=pick(1, 2),
        decoder_layers=pick(1, 2), dropout=0.0, activation_fn=act, encoder_prenorm=prenorm, decoder_prenorm=prenorm,
        encoder_positional_encoding=pos, decoder_positional_encoding=pos,
        rms_norm=bool(rs.randint(2)), has_bias=bool(rs.randint(2)), norm_bias=bool(rs.randint(2)),
        attention_key_bias=bool(rs.randint(2)), scale_attn=bool(rs.randint(4) > 0), scale_embed=bool(rs.randint(2)),
        shared_embeddings=bool(rs.randint(2)), tied_output_projection=bool(rs.randint(2)),
        shared_norm=prenorm and bool(rs.randint(2)), encoder_embed_norm=bool(rs.randint(2)),
        decoder_embed_norm=bool(rs.randint(2)), label_smoothing=pick(0.0, 0.1), prompt_loss=pick(1.0, 1.0, 0.5, 0.0),
        encoder_max_len=64, decoder_max_len=64, norm_eps=pick(1e-5, 1e-6))
    V = int(pick(37, 64, 101))
    model = Transformer(cfg, DistributedConfig(), SyntheticTask(V))
    load_paramgen(model, 400 + seed)
    names_shapes = [(k, tuple(v.shape)) for k, v in model.state_dict().items()]
    P = {k: v.requires_grad_() for k, v in O.to_torch_state(paramgen.make_state_dict(400 + seed, names_shapes)).items()}
    first = {}
    for k, v in model.state_dict().items():  # tied tensors carry the value of their first alias (as load_paramgen does)
        a = first.setdefault(v.data_ptr(), k)
        if a != k:
            P[k] = P[a]
    model = model.cuda().train()
    B, S, T = int(rs.randint(1, 5)), int(rs.randint(2, 20)), int(rs.randint(2, 14))
    batch = paramgen.make_text_batch(500 + seed, B, S, T, V, prompt_cols=int(rs.randint(0, 3)))
    tb = {k: torch.from_numpy(v) for k, v in batch.items()}
    ref_loss, ref_logs = O.transformer_forward(P, cfg, **tb)
    ref_loss.backward()
    loss, logs = model(**{k: v.cuda() for k, v in tb.items()})
    loss.backward()
    what = {k: getattr(cfg, k) for k in ('embed_dim', 'activation_fn', 'encoder_prenorm', 'encoder_positional_encoding',
                                         'rms_norm', 'has_bias', 'norm_bias', 'shared_norm', 'shared_embeddings',
                                         'tied_output_projection', 'prompt_loss', 'label_smoothing', 'scale_attn')}
    assert abs(loss.item() - ref_loss.item()) <= 1e-4 * abs(ref_loss.item()), what
    assert logs['num_tokens'] == ref_logs['num_tokens'], what
    for k in ref_logs:
        assert abs(logs[k] - ref_logs[k]) <= 1e-4 * abs(ref_logs[k]) + 1e-9, (k, what)
    gmax = max(v.grad.abs().max().item() for v in P.values() if v.grad is not None)
    for n, p in model.named_parameters():
        want = P[n].grad
        if want is None:  # (e.g. a parameter the configuration never reaches)
            assert p.grad is None or p.grad.abs().max().item() == 0, (n, what)
            continue
        assert p.grad is not None, (n, what)
        # relative to the tensor's own scale, with a floor tied to the largest gradient of the model: with one or two
        # target positions the softmax is (nearly) constant and dK is what is left of a cancellation
        err = (p.grad.double().cpu() - want.double()).abs().max().item()
        assert err <= 3e-4 * want.abs().max().item() + 1e-5 * gmax, (n, err, want.abs().max().item(), gmax, what)
