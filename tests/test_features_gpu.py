"""Feature collate on the device (pasero_amd/features.py, pk_pad_rows) against the reference's CPU semantics
(utils.tokens_as_tensor, pasero/utils.py:709-736: pad_sequence(batch_first, 0.0) then `.to(dtype)`): bit-exact."""
import numpy as np
import pytest
import torch
from torch.nn.utils.rnn import pad_sequence

pytestmark = pytest.mark.gpu


def _reference(token_list, dtype):
    """the reference's tokens_as_tensor for floating-point inputs, restated with the same torch calls"""
    seqs = [torch.as_tensor(t) for t in token_list]
    tokens = pad_sequence(seqs, batch_first=True, padding_value=0.0).to(dtype)
    return tokens, torch.LongTensor([len(t) for t in token_list])


@pytest.mark.parametrize('src', [np.float16, np.float32])
@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
@pytest.mark.parametrize('lens,D', [((300, 7, 150), 80), ((1,), 1024), ((5, 5), 3), ((0, 9, 2), 80), ((1000, 999, 37, 512), 1024)])
def test_collate_features_matches_reference(src, dtype, lens, D):
    if not torch.cuda.is_available():
        pytest.skip('needs a GPU')
    from pasero_amd.features import collate_features
    rs = np.random.RandomState(3)
    arrays = [(rs.standard_normal((n, D)) * 3).astype(src) for n in lens]
    got, got_len = collate_features(arrays, dtype, 'cuda')
    want, want_len = _reference(arrays, dtype)
    assert got.shape == want.shape and got.dtype == dtype
    assert torch.equal(got.cpu(), want) and torch.equal(got_len, want_len)


@pytest.mark.parametrize('in_memory', [True, False])
def test_collate_from_feature_file_matches_the_reference(in_memory, tmp_path):
    """file written by the real NumpyFile.build -> rows -> the real utils.tokens_as_tensor (golden: features_file):
    bit-exact in fp32 and bf16, lengths included, for three batches (one of them re-reads rows and holds an empty clip)"""
    from conftest import load_golden
    from pasero_amd.features import NumpyFile, collate_from_file
    g = load_golden('features_file')
    data = g['file_bytes'].tobytes()
    if in_memory:
        f = NumpyFile(data)
    else:
        (tmp_path / 'f.bin').write_bytes(data)
        f = NumpyFile(str(tmp_path / 'f.bin'))
    for bi, spec in enumerate(g['batches']):
        b = [int(x) for x in str(spec).split(',')]
        for name, dt in (('f32', torch.float32), ('bf16', torch.bfloat16)):
            got, lens = collate_from_file(f, b, dt, 'cuda')
            want = torch.from_numpy(g[f'batch{bi}_{name}'])
            if dt == torch.bfloat16:
                want = want.view(torch.bfloat16)
            assert got.dtype == dt and got.shape == want.shape, (bi, name)
            assert torch.equal(got.cpu(), want), (bi, name)
            assert lens.dtype == torch.int64 and torch.equal(lens, torch.from_numpy(g[f'batch{bi}_len']))


def test_wav_to_log_mel_matches_oracle():
    if not torch.cuda.is_available():
        pytest.skip('needs a GPU')
    from oracle import ref_cpu as O
    from pasero_amd.features import wav_to_log_mel
    rs = np.random.RandomState(8)
    wavs = [0.1 * rs.standard_normal(16000 * 3).astype(np.float32), 0.05 * rs.standard_normal(40000).astype(np.float32)]
    feats, lens = wav_to_log_mel(wavs, 'cuda')
    assert feats.shape == (2, 3000, 80) and lens.tolist() == [3000, 3000]
    for i, w in enumerate(wavs):
        assert np.abs(feats[i].cpu().numpy() - O.log_mel(w)).max() < 5e-4  # tolerance of the K8 kernel tests
