import os
import sys
import json
import types
import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, 'tests', 'golden')
for p in (ROOT, GOLDEN):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with `-m gpu` through gpurun)')


def load_golden(name):
    return np.load(os.path.join(GOLDEN, name + '.npz'), allow_pickle=False)


def golden_cfg(g):
    """cfg stored as JSON in a fixture -> attribute namespace (duck-types pasero.config.TransformerConfig)"""
    return types.SimpleNamespace(**json.loads(str(g['cfg'])))


def golden_names_shapes(g, prefix=''):
    names = [str(n) for n in g[prefix + 'param_names']]
    shapes = [tuple(int(x) for x in str(s).split(',')) if str(s) else () for s in g[prefix + 'param_shapes']]
    return list(zip(names, shapes))


@pytest.fixture(scope='session')
def golden():
    return load_golden
