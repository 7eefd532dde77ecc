"""The development knobs of the LDS-DMA kernels select other instantiations of the same templates (512-pixel tiles, single buffer,
4 / 8-channel chunks, wave priorities, the register-staged kernels).  They are not defaults, but they are compiled in: each must
still give the right numbers (tools/knob_check.py: plain torch on the CPU as the reference, batch 64, NaN-poisoned LDS)."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

# The Winograd kernels (TG_CONV_WINO, default 1) take the stride-1 3x3 launches that fill the chip and the >= 32-channel pooled
# convolutions; the direct kernels behind them -- what every TG_DMA_* knob selects among -- run with TG_CONV_WINO=0.
_DIRECT = [{'TG_DMA_TILE': '512'}, {'TG_DMA_TILE': '512', 'TG_DMA_CK': '4'}, {'TG_DMA_DB': '0'}, {'TG_DMA_CK': '4'}, {'TG_DMA_CK': '8'},
           {'TG_DMA_PRIO': '1'}, {'TG_DMA_WGRAD_DB': '1'}, {'TG_DMA_S2': '0'}, {'TG_DMA_WGRAD': '0'}, {'TG_DMA_KSPLIT': '0'}, {'TG_CONV_DMA': '0'}]
KNOBS = [{}, {'TG_CONV_WINO': '0'}, {'TG_CONV_WINO': '2'}] + [dict(k, TG_CONV_WINO='0') for k in _DIRECT]


@pytest.mark.parametrize('knobs', KNOBS, ids=lambda k: ','.join(f'{a[3:]}={b}' for a, b in k.items()) or 'defaults')
def test_kernels_under_knob(knobs):
    env = dict(os.environ, **knobs)
    r = subprocess.run([sys.executable, os.path.join(REPO, 'tools', 'knob_check.py')], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
