"""CPU oracle: a functional restatement of tartangan's SA-GAN / SA-GAN-IQN
G+D training step in plain PyTorch CPU fp32.

TEST INFRASTRUCTURE ONLY.  Only ``tests/``, ``__graft_entry__.smoke()`` and
``bench.py``'s ``cpu_baseline`` leg may import this module; the product path
(``tartangan_amd``) never does and fails loudly without its HIP library.

Parity status: PINNED.  Every function here is checked against golden
fixtures produced by running the reference's own code
(``tests/golden/make_golden.py`` -> ``tests/golden/*.json``) by
``tests/test_oracle_golden.py``.  The arithmetic itself lives in PyTorch
(third-party, reference pin torch>=1.4,<=1.5, here 2.10.0); the oracle calls
the same ATen ops in the same order as the reference's modules.

State is held as flat ``{state_dict key: tensor}`` dicts using exactly the
reference's key names, so the same procedural weights load
into the reference modules, this oracle and the HIP modules.

Reference map (all paths relative to /root/reference/tartangan):
  g_forward          models/pluggan.py:58-84, models/blocks/generator.py:32-80,115-129
  d_forward          models/pluggan.py:88-132, models/blocks/discriminator.py:11-22,49-95,126-178
  self_attention     models/blocks/attention.py:21-35
  iqn_head/iqn_loss  models/iqn.py:27-46,76-130
  gradient_penalty   models/losses.py:17-30
  OracleTrainer      trainers/cnn.py:29-165, trainers/iqn.py:29-156, trainers/trainer.py:153-176
"""
import math
from collections import OrderedDict, namedtuple

import numpy as np
import torch
import torch.nn.functional as F

GANConfig = namedtuple('GANConfig', 'base_size latent_dims data_dims blocks attention')

# models/pluggan.py:199-249 (the configs BASELINE.json names)
CONFIGS = {
    '16': GANConfig(4, 100, 3, (64, 32), ()),
    '32': GANConfig(4, 128, 3, (128, 64, 32), ()),
    '64': GANConfig(4, 128, 3, (128, 128, 64, 32), ()),
    '128': GANConfig(4, 256, 3, (128, 128, 64, 32, 16), ()),
}

SLOPE = 0.2
BN_EPS = 1e-5
BN_MOMENTUM = 0.1
NUM_QUANTILES = 8      # models/iqn.py:78
QUANTILE_DIMS = 20     # models/iqn.py:78


# option variants of the trainers (trainers/cnn.py:32-45): --norm bn|id, --activation relu|selu|elu, --g-base mlp|tiledz.
# Module-level switches set by OracleTrainer (the oracle is single-threaded test code).
class Options:
    norm = 'bn'
    activation = 'relu'
    g_base = 'mlp'


def act(x):
    """activation_factory() of trainers/cnn.py:41-45"""
    if Options.activation == 'relu':
        return F.leaky_relu(x, SLOPE)
    if Options.activation == 'selu':
        return F.selu(x)
    if Options.activation == 'elu':
        return F.elu(x)
    raise KeyError(Options.activation)


def get_config(name, attention=None, blocks=None, latent_dims=None, base_size=None):
    """A named config, or (fixtures of configs this table does not hold / scaled models) explicit widths."""
    if name in CONFIGS:
        cfg = CONFIGS[name]
    else:
        cfg = GANConfig(base_size or 4, latent_dims, 3, tuple(blocks), ())
    if blocks is not None:
        cfg = cfg._replace(blocks=tuple(blocks))          # GANConfig.scale_model (models/pluggan.py:24-28)
    if latent_dims is not None:
        cfg = cfg._replace(latent_dims=latent_dims)
    if attention is not None:
        cfg = cfg._replace(attention=tuple(attention))
    return cfg


# --------------------------------------------------------------------------
# state templates (key order == reference state_dict order)
# --------------------------------------------------------------------------
def _conv_t(S, key, cout, cin, k, bias=True):
    S[key + '.weight'] = torch.zeros(cout, cin, k, k)
    if bias:
        S[key + '.bias'] = torch.zeros(cout)


def _bn_t(S, key, c):
    if Options.norm == 'id':          # nn.Identity: no parameters, no buffers
        return
    S[key + '.weight'] = torch.ones(c)
    S[key + '.bias'] = torch.zeros(c)
    S[key + '.running_mean'] = torch.zeros(c)
    S[key + '.running_var'] = torch.ones(c)
    S[key + '.num_batches_tracked'] = torch.zeros((), dtype=torch.long)


def _attn_t(S, key, c):
    S[key + '.gamma'] = torch.zeros(())
    _conv_t(S, key + '.theta', c // 8, c, 1, bias=False)
    _conv_t(S, key + '.phi', c // 8, c, 1, bias=False)
    _conv_t(S, key + '.g', c // 2, c, 1, bias=False)
    _conv_t(S, key + '.o', c, c // 2, 1, bias=False)


def g_template(cfg):
    S = OrderedDict()
    c0 = cfg.blocks[0]
    if Options.g_base == 'mlp':       # TiledZGeneratorInput (generator.py:101-112) has no parameters
        S['blocks.0.base_img.0.weight'] = torch.zeros(cfg.base_size ** 2 * c0, cfg.latent_dims)
        S['blocks.0.base_img.0.bias'] = torch.zeros(cfg.base_size ** 2 * c0)
    bi, cin = 1, c0
    for i, cout in enumerate(cfg.blocks):
        p = f'blocks.{bi}'
        if cin != cout:
            _conv_t(S, p + '.project_input.0', cout, cin, 1)
        if i == 0:
            _conv_t(S, p + '.convs.0', cout, cin, 3)
            _bn_t(S, p + '.convs.1', cout)
            _conv_t(S, p + '.convs.3', cout, cout, 3)
        else:
            _bn_t(S, p + '.convs.0', cin)
            _conv_t(S, p + '.convs.2', cout, cin, 3)
            _bn_t(S, p + '.convs.3', cout)
            _conv_t(S, p + '.convs.5', cout, cout, 3)
        bi += 1
        if i in cfg.attention:
            _attn_t(S, f'blocks.{bi}', cout)
            bi += 1
        cin = cout
    p = f'blocks.{bi}'
    _bn_t(S, p + '.convs.0', cin)
    _conv_t(S, p + '.convs.2', cfg.data_dims, cin, 1)
    return S


def d_template(cfg, iqn=False):
    S = OrderedDict()
    c_last = cfg.blocks[-1]
    rev = list(reversed(list(enumerate(cfg.blocks))))
    c_top = rev[-1][1]
    if iqn:
        _bn_t(S, 'to_output.activation.0', c_top)
        S['to_output.to_output.0.weight'] = torch.zeros(1, c_top)
        S['to_output.to_output.0.bias'] = torch.zeros(1)
        S['to_output.iqn.quantile_embedding.embedding_range'] = torch.arange(1, QUANTILE_DIMS + 1).float()
        S['to_output.iqn.quantile_embedding.to_state.0.weight'] = torch.zeros(c_top, QUANTILE_DIMS)
        S['to_output.iqn.quantile_embedding.to_state.0.bias'] = torch.zeros(c_top)
        bi, cin, first = 0, cfg.data_dims, False
    else:
        _conv_t(S, 'blocks.0.convs.0', c_last, cfg.data_dims, 1)
        bi, cin, first = 1, c_last, True
    for i, cout in rev:
        p = f'blocks.{bi}'
        if first:
            _conv_t(S, p + '.convs.0', cout, cin, 3)
            _bn_t(S, p + '.convs.1', cout)
            _conv_t(S, p + '.convs.3', cout, cout, 3)
        else:
            _bn_t(S, p + '.convs.0', cin)
            _conv_t(S, p + '.convs.2', cout, cin, 3)
            _bn_t(S, p + '.convs.3', cout)
            _conv_t(S, p + '.convs.5', cout, cout, 3)
        if cin != cout:
            _conv_t(S, p + '.project_input.0', cout, cin, 1)
        first = False
        bi += 1
        if i in cfg.attention:
            _attn_t(S, f'blocks.{bi}', cout)
            bi += 1
        cin = cout
    if not iqn:
        p = f'blocks.{bi}'
        _bn_t(S, p + '.activation.0', cin)
        S[p + '.to_output.0.weight'] = torch.zeros(1, cin)
        S[p + '.to_output.0.bias'] = torch.zeros(1)
    return S


def is_param(key):
    leaf = key.rsplit('.', 1)[-1]
    return leaf in ('weight', 'bias', 'gamma')


def default_init_(S, seed_continue=True):
    """nn.Module default initialisation, in the reference's construction order
    (nn.Linear / nn.Conv2d: kaiming_uniform_(a=sqrt(5)) + bias U(-1/sqrt(fan_in),..);
    BatchNorm ones/zeros; gamma 0).  Consumes the global CPU RNG exactly like
    constructing the reference modules does."""
    def rank(item):
        # construction order differs from state_dict order in two places:
        # a block builds its 3x3 convs before project_input (generator.py:37-54)
        # and IQNDiscriminator builds its blocks before to_output (pluggan.py:117-124)
        idx, k = item
        parts = k.split('.')
        block = int(parts[1]) if parts[0] == 'blocks' else 10 ** 6
        return (block, 1 if 'project_input' in k else 0, idx)

    keys = [k for _, k in sorted(enumerate(S.keys()), key=rank)]
    for k in keys:
        t = S[k]
        leaf = k.rsplit('.', 1)[-1]
        if leaf == 'weight' and t.dim() >= 2:
            torch.nn.init.kaiming_uniform_(t, a=math.sqrt(5))
            bkey = k[:-6] + 'bias'
            if bkey in S:
                fan_in = t[0].numel()
                bound = 1 / math.sqrt(fan_in) if fan_in > 0 else 0
                torch.nn.init.uniform_(S[bkey], -bound, bound)
    return S


# --------------------------------------------------------------------------
# layers
# --------------------------------------------------------------------------
def _bn(S, key, x, training):
    if Options.norm == 'id':
        return x
    if training:
        S[key + '.num_batches_tracked'] += 1
    return F.batch_norm(x, S[key + '.running_mean'], S[key + '.running_var'],
                        S[key + '.weight'], S[key + '.bias'], training, BN_MOMENTUM, BN_EPS)


def _conv(S, key, x, pad):
    return F.conv2d(x, S[key + '.weight'], S.get(key + '.bias'), padding=pad)


def _res_convs(S, p, x, first, training):
    if first:
        h = _conv(S, p + '.convs.0', x, 1)
        h = act(_bn(S, p + '.convs.1', h, training))
        return _conv(S, p + '.convs.3', h, 1)
    h = act(_bn(S, p + '.convs.0', x, training))
    h = _conv(S, p + '.convs.2', h, 1)
    h = act(_bn(S, p + '.convs.3', h, training))
    return _conv(S, p + '.convs.5', h, 1)


def self_attention(S, p, x):
    """models/blocks/attention.py:21-35"""
    b, c, hh, ww = x.shape
    n = hh * ww
    theta = F.conv2d(x, S[p + '.theta.weight'])
    phi = F.max_pool2d(F.conv2d(x, S[p + '.phi.weight']), [2, 2])
    g = F.max_pool2d(F.conv2d(x, S[p + '.g.weight']), [2, 2])
    theta = theta.view(-1, c // 8, n)
    phi = phi.view(-1, c // 8, n // 4)
    g = g.view(-1, c // 2, n // 4)
    beta = F.softmax(torch.bmm(theta.transpose(1, 2), phi), -1)
    o = torch.bmm(g, beta.transpose(1, 2)).view(-1, c // 2, hh, ww)
    o = F.conv2d(o, S[p + '.o.weight'])
    return S[p + '.gamma'] * o + x


def g_forward(S, z, cfg, training=True):
    c0 = cfg.blocks[0]
    if Options.g_base == 'tiledz':    # generator.py:110-112
        x = z[..., None, None].repeat(1, 1, cfg.base_size, cfg.base_size)
    else:
        x = act(F.linear(z, S['blocks.0.base_img.0.weight'], S['blocks.0.base_img.0.bias']))
        x = x.view(-1, c0, cfg.base_size, cfg.base_size)
    bi, cin = 1, c0
    for i, cout in enumerate(cfg.blocks):
        p = f'blocks.{bi}'
        x = F.interpolate(x, scale_factor=2, mode='nearest')
        h = _res_convs(S, p, x, i == 0, training)
        if cin != cout:
            x = _conv(S, p + '.project_input.0', x, 0)
        x = x + h
        bi += 1
        if i in cfg.attention:
            x = self_attention(S, f'blocks.{bi}', x)
            bi += 1
        cin = cout
    p = f'blocks.{bi}'
    x = act(_bn(S, p + '.convs.0', x, training))
    return torch.tanh(_conv(S, p + '.convs.2', x, 0))


def _d_trunk(S, x, cfg, iqn, training):
    rev = list(reversed(list(enumerate(cfg.blocks))))
    if iqn:
        bi, cin, first = 0, cfg.data_dims, False
    else:
        x = _conv(S, 'blocks.0.convs.0', x, 0)
        bi, cin, first = 1, cfg.blocks[-1], True
    for i, cout in rev:
        p = f'blocks.{bi}'
        h = F.avg_pool2d(_res_convs(S, p, x, first, training), 2)
        x = F.interpolate(x, scale_factor=0.5, mode='bilinear', align_corners=True)
        if cin != cout:
            x = _conv(S, p + '.project_input.0', x, 0)
        x = x + h
        first = False
        bi += 1
        if i in cfg.attention:
            x = self_attention(S, f'blocks.{bi}', x)
            bi += 1
        cin = cout
    return x, bi


def d_forward(S, x, cfg, training=True):
    x, bi = _d_trunk(S, x, cfg, False, training)
    p = f'blocks.{bi}'
    x = act(_bn(S, p + '.activation.0', x, training))
    x = torch.sum(x, [2, 3])
    return F.linear(x, S[p + '.to_output.0.weight'], S[p + '.to_output.0.bias'])


def sample_taus(batch):
    """models/iqn.py:105-108 -- CPU default generator, (8B,1), row = q*B + b."""
    return torch.rand(batch * NUM_QUANTILES, 1)


def cosine_embedding(S, p, taus):
    """models/iqn.py:41-46; note the evaluation order (qs*pi)*range in fp32."""
    qs = taus.repeat(1, QUANTILE_DIMS)
    qs = qs * np.pi * S[p + '.embedding_range']
    qs = torch.cos(qs)
    return torch.tanh(F.linear(qs, S[p + '.to_state.0.weight'], S[p + '.to_state.0.bias']))


def iqn_loss(preds, target, taus, k=1.):
    """models/iqn.py:111-130"""
    batch = target.shape[0]
    out_dims = target.shape[-1]
    nq = preds.shape[0] // batch
    taus = taus.reshape(-1, batch, out_dims)
    preds = preds.reshape(-1, batch, out_dims)
    target = target.repeat(nq, 1).reshape(-1, batch, out_dims)
    err = target - preds
    loss = torch.where(err.abs() <= k, 0.5 * err.pow(2), k * (err.abs() - 0.5 * k))
    return ((taus - (err < 0).float()).abs() * loss).sum(0).mean()


def iqn_d_forward(S, x, cfg, targets=None, training=True, taus=None):
    x, _ = _d_trunk(S, x, cfg, True, training)
    p = 'to_output'
    feats = act(_bn(S, p + '.activation.0', x, training))
    feats = torch.sum(feats, [2, 3])
    batch = feats.shape[0]
    feats_rep = feats.repeat(NUM_QUANTILES, 1)
    if taus is None:
        taus = sample_taus(batch).to(feats.device)
    emb = cosine_embedding(S, p + '.iqn.quantile_embedding', taus)
    feats_tau = feats_rep * emb
    p_tau = F.linear(feats_tau, S[p + '.to_output.0.weight'], S[p + '.to_output.0.bias'])
    loss = None
    if targets is not None:
        loss = iqn_loss(p_tau, targets, taus.repeat(1, 1))
    p_target = p_tau.reshape(NUM_QUANTILES, -1, 1).mean(0)
    if targets is not None:
        return p_target, loss
    return p_target


def gradient_penalty(preds, data):
    """models/losses.py:17-30 (R1)"""
    batch = data.size(0)
    grad = torch.autograd.grad(outputs=preds.sum(), inputs=data, create_graph=True,
                               retain_graph=True, only_inputs=True)[0]
    return grad.pow(2).view(batch, -1).sum(1).mean()


# --------------------------------------------------------------------------
# the training step
# --------------------------------------------------------------------------
class OracleTrainer:
    """trainers/cnn.py:29-165 and trainers/iqn.py:29-156 restated on state dicts."""

    def __init__(self, config, kind='cnn', batch_size=64, attention=None, lr_g=1e-4, lr_d=4e-4,
                 lr_target_g=1e-3, grad_penalty=5., device='cpu', norm='bn', activation='relu', g_base='mlp',
                 model_scale=1., blocks=None, latent_dims=None):
        Options.norm, Options.activation, Options.g_base = norm, activation, g_base
        if kind == 'iqn' and activation == 'elu':
            raise KeyError(activation)                   # trainers/iqn.py:41-44 has no 'elu'
        self.cfg = get_config(config, attention, blocks, latent_dims) if isinstance(config, str) else config
        self.cfg = self.cfg._replace(blocks=tuple(int(c * model_scale) for c in self.cfg.blocks))
        self.kind = kind
        self.batch_size = batch_size
        self.lr_target_g = lr_target_g
        self.grad_penalty = grad_penalty
        self.device = device
        # construction order g, target_g, d (cnn.py:66-83) consumes the init RNG
        self.g = default_init_(g_template(self.cfg))
        self.target_g = default_init_(g_template(self.cfg))
        self.d = default_init_(d_template(self.cfg, iqn=(kind == 'iqn')))
        self._make_optimizers(lr_g, lr_d)
        if activation == 'selu':                 # cnn.py:92-94
            self._init_params_selu(self.g)
            self._init_params_selu(self.d)
        self.update_target_generator()       # cnn.py:95 (lr argument is ignored there)

    def _init_params_selu(self, S):
        """cnn.py:97-105: vectors zeroed, everything else N(0, 1/fan_in), in .parameters() order."""
        with torch.no_grad():
            for p in self._params(S):
                if p.dim() == 1:
                    p.zero_()
                else:
                    fan_in, _ = torch.nn.init._calculate_fan_in_and_fan_out(p)
                    p.normal_(std=float(np.sqrt(1. / fan_in)))

    def _params(self, S):
        return [v for k, v in S.items() if is_param(k)]

    def _make_optimizers(self, lr_g, lr_d):
        self.lr_g, self.lr_d = lr_g, lr_d
        self.opt_g = torch.optim.Adam(self._params(self.g), lr=lr_g, betas=(0., 0.999))
        self.opt_d = torch.optim.Adam(self._params(self.d), lr=lr_d, betas=(0., 0.999))

    def load(self, g=None, target_g=None, d=None):
        for S, new in ((self.g, g), (self.target_g, target_g), (self.d, d)):
            if new is None:
                continue
            assert list(S.keys()) == list(new.keys()), 'state_dict keys differ'
            with torch.no_grad():
                for k in S:
                    S[k].copy_(new[k])

    @staticmethod
    def _toggle(S, on):
        for k, v in S.items():
            if is_param(k):
                v.requires_grad_(on)

    def sample_z(self, n=None):
        return torch.randn(n or self.batch_size, self.cfg.latent_dims)

    def _d(self, x, targets=None):
        if self.kind == 'iqn':
            return iqn_d_forward(self.d, x, self.cfg, targets=targets)
        return d_forward(self.d, x, self.cfg)

    def train_batch(self, imgs):
        B = self.batch_size
        # ---- D phase (cnn.py:112-137 / iqn.py:108-130)
        self._toggle(self.g, False)
        self._toggle(self.d, True)
        self.opt_d.zero_grad()
        fake = g_forward(self.g, self.sample_z(len(imgs)), self.cfg)
        labels = torch.zeros(2 * len(imgs), 1)
        labels[:len(labels) // 2] = 1
        real = imgs.clone().requires_grad_() if self.grad_penalty else imgs
        if self.kind == 'iqn':
            p_real, l_real = self._d(real, labels[:B])
            p_fake, l_fake = self._d(fake.detach(), labels[B:])
            d_loss = l_real + l_fake
        else:
            p_real = self._d(real)
            p_fake = self._d(fake.detach())
            d_loss = F.binary_cross_entropy_with_logits(torch.cat([p_real, p_fake], 0), labels)
        gp = 0.
        if self.grad_penalty:
            gp = self.grad_penalty * gradient_penalty(p_real, real)
            d_loss = d_loss + gp
        d_loss.backward()
        self.opt_d.step()
        # ---- G phase (cnn.py:139-149 / iqn.py:132-140)
        self._toggle(self.g, True)
        self._toggle(self.d, False)
        self.opt_g.zero_grad()
        fake = g_forward(self.g, self.sample_z(len(imgs)), self.cfg)
        ones = torch.ones(len(fake), 1)
        if self.kind == 'iqn':
            _, g_loss = self._d(fake, ones)
        else:
            g_loss = F.binary_cross_entropy_with_logits(self._d(fake), ones)
        g_loss.backward()
        self.opt_g.step()
        self.update_target_generator()
        return dict(g_loss=float(g_loss.detach()), d_loss=float(d_loss.detach()),
                    gp=float(gp.detach() if torch.is_tensor(gp) else gp))

    @torch.no_grad()
    def update_target_generator(self):
        """cnn.py:158-165: always lr_target_g, whatever ``lr`` was passed."""
        for gp_, tp in zip(self._params(self.g), self._params(self.target_g)):
            tp.add_((gp_ - tp) * self.lr_target_g)
