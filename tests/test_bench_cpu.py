"""bench.py's launcher contract on CPU: `python bench.py --gpus N` with no launcher environment starts its own N ranks
(one process per GPU, like the reference's pasero-train, cli/train.py:705-727), rank 0's single JSON line comes through
and the parent exits with the children's return code.  `--rehearse-cpu` swaps the HIP model (no CPU path) for a small
torch MLP over gloo, so what runs here is the real plumbing: self-launch, rendezvous, bucketed reducer, fused logs
all-reduce, barrier + max-over-ranks timing, JSON."""
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT


def _run(*extra, env=None):
    e = dict(os.environ)
    for k in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_ADDR', 'MASTER_PORT'):
        e.pop(k, None)
    e.update(env or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--rehearse-cpu', '--steps', '3',
                           '--warmup', '1', *extra], capture_output=True, text=True, timeout=300, env=e)


def _json_lines(out: str):
    return [json.loads(line) for line in out.splitlines() if line.startswith('{')]


@pytest.mark.timeout(400)
def test_bench_self_launches_two_ranks():
    r = _run('--gpus', '2')
    assert r.returncode == 0, r.stderr[-2000:]
    lines = _json_lines(r.stdout)
    assert len(lines) == 1, r.stdout
    j = lines[0]
    assert j['n_gpus'] == 2 and j['rccl_ranks'] == 2 and j['steps'] == 3 and j['scaling'] == 'weak'
    assert j['value'] > 0 and 'rehearsal' in j
    # whole-job tokens: both ranks' tokens are in `value`
    assert abs(j['value'] * j['ms_per_step'] * 1e-3 - 2 * j['config']['tokens_per_rank_per_step']) < 1e-6
    # the reducer explains itself in the line: transport, buckets in launch order and their sizes
    ar = j['config']['gradient_all_reduce']
    assert ar['world_size'] == 2 and 'gloo' in ar['transport'] and ar['schedule_trial'] is None
    assert ar['buckets'] == len(ar['bucket_mib']) >= 2 and abs(sum(ar['bucket_mib']) - ar['total_mib']) < 1e-2
    assert ar['last_bucket_mib'] == ar['bucket_mib'][-1] and ar['largest_bucket_mib'] == max(ar['bucket_mib'])


def test_bench_single_rank_is_not_relaunched():
    r = _run('--gpus', '1')
    assert r.returncode == 0, r.stderr[-2000:]
    lines = _json_lines(r.stdout)
    assert len(lines) == 1 and lines[0]['n_gpus'] == 1


def test_bench_under_an_external_launcher_checks_the_world_size():
    r = _run('--gpus', '2', env={'WORLD_SIZE': '1', 'RANK': '0'})
    assert r.returncode != 0 and 'WORLD_SIZE=1' in (r.stderr + r.stdout)
