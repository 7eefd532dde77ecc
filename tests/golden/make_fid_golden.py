#!/usr/bin/env python3
"""Golden vectors for the FID / Inception-score math: runs the REFERENCE's own functions
(tartangan/inception_utils.py: torch_cov, sqrt_newton_schulz, torch_calculate_frechet_distance,
calculate_inception_score) on procedural features and writes tests/golden/fid_math.json (numbers only).
Build container only (needs /root/reference); torchvision / smart_open are absent and never called by these
functions -- they are registered as inert modules exactly as in make_golden.py."""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden  # noqa: F401,E402  (installs the inert stand-ins and sys.path for the reference)
import torch  # noqa: E402

from oracle.procedural import summarize  # noqa: E402
from oracle.fid_features import procedural_features, procedural_probs  # noqa: E402
from tartangan import inception_utils as R  # noqa: E402

CASES = {
    # name: (D, N, classes, splits)
    'd64_n200': (64, 200, 50, 5),
    'd256_n1000': (256, 1000, 100, 10),
    'd2048_n1000': (2048, 1000, 1000, 5),        # config 5: --n-inception-imgs 1000 (metrics/fid.py:52), num_splits=5 (:40)
}


def run(name):
    D, N, classes, splits = CASES[name]
    gen, data = procedural_features(N, D, 11), procedural_features(N, D, 12, shift=0.25)
    out = dict(D=D, N=N, classes=classes, splits=splits)
    mu1, mu2 = torch.mean(gen, 0), torch.mean(data, 0)
    s1 = R.torch_cov(gen.clone(), rowvar=False)
    s2 = R.torch_cov(data.clone(), rowvar=False)
    out['cov'] = summarize(s1, 8)
    out['cov_trace'] = float(torch.trace(s1))
    g2 = gen.clone()
    R.torch_cov(g2, rowvar=False)
    out['centred_in_place'] = summarize(g2, 4)       # the reference centres its argument: pin the side effect too
    root = R.sqrt_newton_schulz(s1.mm(s2).unsqueeze(0), 20).squeeze()
    out['sqrt'] = summarize(root, 8)
    out['sqrt_trace'] = float(torch.trace(root))
    out['fid'] = float(R.torch_calculate_frechet_distance(mu1, s1, mu2, s2))
    probs = procedural_probs(N, classes, 13)
    m, s = R.calculate_inception_score(probs.numpy(), splits)
    out['is_mean'], out['is_std'] = float(m), float(s)
    print(name, {k: v for k, v in out.items() if not isinstance(v, dict)}, flush=True)
    return out


def run_frontend():
    """The input side: the reference's own accumulate_inception_activations + WrapInception (normalise twice, resize to
    299 x 299, run the wrapped network's layers, softmax) around a stand-in network with torchvision's attribute names."""
    from oracle.fid_features import blocky_images, tiny_inception
    D, classes, batch, size, want = 64, 10, 8, 32, 40
    inner = tiny_inception(D, classes, 5)
    net = R.WrapInception(inner)
    seen = []
    inner.Conv2d_1a_3x3.register_forward_pre_hook(lambda m, inp: seen.append(inp[0].detach().clone()))
    calls = [0]

    def sample():
        calls[0] += 1
        return blocky_images(batch, size, 700 + calls[0])
    pool, probs = R.accumulate_inception_activations(sample, net, want)
    out = dict(D=D, classes=classes, batch=batch, size=size, want=want, calls=calls[0], n=int(pool.shape[0]),
               preprocessed_first_batch=summarize(seen[0], 16), pool=summarize(pool, 16), probs=summarize(probs, 16))
    m, s = R.calculate_inception_score(probs.numpy(), 5)
    data = procedural_features(200, D, 12, shift=0.25)
    mu2, s2 = torch.mean(data, 0), R.torch_cov(data.clone(), rowvar=False)
    mu1, s1 = torch.mean(pool, 0), R.torch_cov(pool.clone(), rowvar=False)
    out.update(is_mean=float(m), is_std=float(s), fid=float(R.torch_calculate_frechet_distance(mu1, s1, mu2, s2)),
               data_features=dict(n=200, seed=12, shift=0.25))
    # a non-square, non-multiple-of-anything source size through WrapInception alone (one normalisation)
    x = blocky_images(3, 40, 901)[:, :, :, :27].contiguous()
    seen.clear()
    net(x)
    out['wrap_only_40x27'] = summarize(seen[0], 16)
    print('frontend', {k: v for k, v in out.items() if not isinstance(v, dict)}, flush=True)
    return out


if __name__ == '__main__':
    torch.set_num_threads(8)
    if sys.argv[1:] == ['frontend']:
        path = os.path.join(HERE, 'fid_frontend.json')
        json.dump(run_frontend(), open(path, 'w'), separators=(',', ':'))
        print('wrote', path)
        sys.exit(0)
    res = {n: run(n) for n in (sys.argv[1:] or CASES)}
    path = os.path.join(HERE, 'fid_math.json')
    old = json.load(open(path)) if os.path.exists(path) else {}
    old.update(res)
    json.dump(old, open(path, 'w'), separators=(',', ':'))
    print('wrote', path)
