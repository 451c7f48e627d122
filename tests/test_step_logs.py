"""`logs` of a step whose numbers are read when first used (pasero_amd.transformer.StepLogs; the reference returns a
plain dict after three .item() syncs, pasero/models/transformer.py:375-380): every way the Trainer and its metrics read a
logs dict (training.py:429-446, utils.gather_dict, json logging) must see the final numbers."""
import json
import pickle

import pytest
import torch

from pasero_amd import transformer as tr


class _Event:
    waited = 0

    def synchronize(self):
        _Event.waited += 1


def _pending(batch_size=4):
    logs = tr.StepLogs.__new__(tr.StepLogs)
    dict.__init__(logs, loss=None, nll_loss=None, num_tokens=None, num_lines=batch_size)
    logs._pending = (torch.tensor([2.0, 1.0, 7.0]), _Event())
    return logs


WANT = {'loss': 2.0 / tr.LN2, 'nll_loss': 1.0 / tr.LN2, 'num_tokens': 7, 'num_lines': 4}


def test_keys_need_no_wait_and_every_read_waits_once():
    _Event.waited = 0
    logs = _pending()
    assert list(logs) == list(WANT) and len(logs) == 4 and 'num_tokens' in logs and list(logs.keys()) == list(WANT)
    assert _Event.waited == 0
    assert logs['num_tokens'] == 7 and isinstance(logs['num_tokens'], int)
    assert logs['loss'] == WANT['loss'] and logs.get('nll_loss') == WANT['nll_loss']
    assert _Event.waited == 1


@pytest.mark.parametrize('read', [
    lambda l: dict(l), lambda l: {**l}, lambda l: dict(l.items()), lambda l: dict(zip(l.keys(), l.values())),
    lambda l: l.copy(), lambda l: json.loads(json.dumps(l)), lambda l: pickle.loads(pickle.dumps(l)),
    lambda l: {k: l[k] for k in l}, lambda l: l | {}, lambda l: {} | l, lambda l: (lambda **kw: kw)(**l),
    lambda l: eval(repr(l)),
])
def test_every_way_of_reading_sees_the_numbers(read):
    got = read(_pending())
    assert got == WANT and type(got) is dict


def test_updates_survive_the_wait_and_equality_compares_values():
    logs = _pending()
    logs['prompt_nll_loss'] = 0.5          # Transformer.forward adds entries before anything is read (:300-321)
    assert logs == {**WANT, 'prompt_nll_loss': 0.5} and logs != WANT
    logs = _pending()
    assert logs.pop('num_lines') == 4 and logs.setdefault('loss', 0.0) == WANT['loss']
    # a value written over one of the pending entries BEFORE the first read is what a later read returns
    logs = _pending()
    logs['loss'] = 9.0
    assert logs['loss'] == 9.0 and logs['num_tokens'] == 7
    logs = _pending()
    logs.update(loss=8.0, extra=1)
    assert dict(logs) == {**WANT, 'loss': 8.0, 'extra': 1}
    logs = _pending()
    logs |= {'nll_loss': 3.0}
    assert logs['nll_loss'] == 3.0 and logs['loss'] == WANT['loss']
    logs = _pending()
    del logs['loss']
    assert 'loss' not in logs and logs['num_tokens'] == 7


@pytest.mark.gpu
def test_model_logs_are_deferred_and_equal_the_eager_ones(monkeypatch):
    if not torch.cuda.is_available():
        pytest.skip('needs a GPU')
    from conftest import load_golden
    from model_utils import build_model, text_batch
    g = load_golden('tiny_encdec_post')
    cfg, model = build_model(g, torch.float32, 'cuda')
    batch = text_batch(g, 'cuda')
    monkeypatch.setattr(tr, '_EAGER_LOGS', True)
    loss0, logs0 = model(**batch)
    assert type(logs0) is dict
    monkeypatch.setattr(tr, '_EAGER_LOGS', False)
    loss1, logs1 = model(**batch)
    assert isinstance(logs1, tr.StepLogs) and logs1._pending is not None
    loss1.backward()                        # enqueued without a host wait
    assert logs1._pending is not None
    assert logs1 == logs0 and logs1._pending is None and loss0.item() == loss1.item()
