"""The driver's round-end entry point on the GPU box: `__graft_entry__.smoke()` runs in a fresh interpreter, loads the in-tree HIP library
(never a fallback) and finds every small call bit-exact against the oracle."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_smoke_runs_in_a_fresh_interpreter(gpu):
    code = ("import __graft_entry__ as g; g.smoke(); "
            "maps = open('/proc/self/maps').read(); "
            "assert 'intel-texture-works-plugin_amd/lib/libispc_texcomp' in maps, 'the in-tree HIP library is not loaded'; print('LOADED')")
    r = subprocess.run([sys.executable, "-c", code], cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:]
    assert "smoke ok: bc7/slow" in r.stdout and "LOADED" in r.stdout, r.stdout[-3000:]
