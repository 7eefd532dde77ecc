"""Winograd F(2x2, 3x3) convolution kernel (csrc/wino.h) beyond the shapes the default dispatch gives it.

By default (TG_CONV_WINO=1) the kernel takes the stride-1 3x3 layers whose launch fills the chip -- those are covered, at the
step's own shapes and batch, by tests/test_kernels_gpu.py (STEP_CONV_SHAPES after LDS poisoning) and by every fixture test.
TG_CONV_WINO=2 hands it EVERY eligible shape: small batches, ragged image groups on the 8x8 / 4x4 geometries (4 / 16 images per
tile), single-workgroup launches, output-channel blocks that are partly masked.  The knob is read once per process, so the
convolution tests run again in a child process with it set."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_conv_tests_with_every_eligible_shape_on_the_winograd_kernel():
    env = dict(os.environ, TG_CONV_WINO='2')
    sel = 'test_conv_fwd or test_conv_dgrad or half_resolution_residual or exact_integer_layout or step_shapes_full_batch'
    r = subprocess.run([sys.executable, '-m', 'pytest', os.path.join(REPO, 'tests', 'test_kernels_gpu.py'), '-x', '-q', '-k', sel,
                        '-p', 'no:cacheprovider'], env=env, cwd=REPO, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    assert r.returncode == 0, r.stdout[-3000:]
    assert ' passed' in r.stdout


def test_winograd_is_at_least_as_accurate_as_the_direct_kernel():
    """Against float64: the transforms add and halve in fp32 but the 16 frequency sums are shorter than the 9 x Cin direct one."""
    code = r'''
import sys, torch, torch.nn.functional as F
sys.path.insert(0, %r)
from tartangan_amd import backend
K = backend.get()
torch.manual_seed(0)
B, Cin, Cout, H = 4, 128, 128, 32
x = torch.randn(B, Cin, H, H); w = torch.randn(Cout, Cin, 3, 3) * (2.0 / (9 * Cin)) ** 0.5; gy = torch.randn(B, Cout, H, H)
y = torch.empty(B, Cout, H, H, device='cuda'); gx = torch.empty(B, Cin, H, H, device='cuda')
K.conv2d_fwd(x.cuda(), w.cuda(), None, None, y, B, Cin, Cout, H, H, 3)
K.conv2d_dgrad(gy.cuda(), w.cuda(), gx, B, Cin, Cout, H, H, 3)
want = F.conv2d(x.double(), w.double(), None, padding=1); wantg = F.conv_transpose2d(gy.double(), w.double(), padding=1)
print(float((y.cpu().double() - want).abs().max() / want.abs().max()), float((gx.cpu().double() - wantg).abs().max() / wantg.abs().max()))
''' % REPO
    errs = {}
    for mode in ('0', '2'):
        r = subprocess.run([sys.executable, '-c', code], env=dict(os.environ, TG_CONV_WINO=mode), stdout=subprocess.PIPE,
                           stderr=subprocess.DEVNULL, text=True, check=True)
        errs[mode] = [float(v) for v in r.stdout.split()[-2:]]
    assert max(errs['2']) < 2e-6, errs
    assert errs['2'][0] <= 1.5 * errs['0'][0] and errs['2'][1] <= 1.5 * errs['0'][1], errs
