"""Drop-in boundary audited against the real reference tree — build container only (the reference never travels).
tools/audit_against_reference.py: configuration fields read, preset defaults, registry resolution, state_dict layouts."""
import os
import subprocess
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(not os.path.isdir('/root/reference/pasero'), reason='reference tree not present on this machine')
def test_boundary_matches_reference_tree():
    env = dict(os.environ, PYTHONDONTWRITEBYTECODE='1')
    r = subprocess.run([sys.executable, os.path.join(REPO, 'tools', 'audit_against_reference.py')], env=env,
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert '0 problem(s)' in r.stdout
