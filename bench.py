#!/usr/bin/env python3
"""Benchmark: images/sec of the full SA-GAN G+D training step (BASELINE.json metric).

  python bench.py --gpus N --steps K --warmup W
  N > 1: either under a launcher (python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...:
  RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* from the environment), or from a bare shell -- the script then starts the N
  ranks itself as fresh child processes (self_launch) and relays rank 0's line.

Workload (config.workload): GAN_CONFIGS['128'] with the author's attention placement (3,)
("128:3"), tartangan.trainers.cnn step semantics (BCE + R1 penalty 5.0, Adam(0,.999) x2, EMA),
fp32, 128x128 RGB, batch 64 PER GPU (weak scaling), synthetic U[-1,1] images resident in HBM,
random-init weights.  One "step" = one train_batch(): D phase incl. second-order R1 backward,
D Adam, G phase, G Adam, EMA, and the reference's loss read-back (one host sync per step).

Rank 0 prints ONE JSON line.  `roofline` is for the dominant kernel family (the MFMA
implicit-GEMM convolution, fwd+dgrad), timed with HIP events on the launch stream in an
instrumented eager pass of the same step right after the timed region; `roofline_step` is
the whole step against the fp32 MFMA floor.  `cpu_baseline` is the CPU oracle
(oracle/sagan_cpu.py, kind "port") timed on this host on a bounded sample.
"""
import argparse
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

FLOP_PER_IMAGE = {'128:3': 1.0446e10, '128': 9.5473e9, '64:1': 7.3881e9, '64': 7.3044e9, '32': 1.6857e9}
MFMA_F32_PEAK_TFLOPS = 157.3      # MI355X_MICROARCH.md, v_mfma_f32_32x32x2_f32 / 16x16x4_f32
HBM_PEAK_GBS = 8000.0


def parse():
    p = argparse.ArgumentParser()
    p.add_argument('--gpus', type=int, default=1)
    p.add_argument('--steps', type=int, default=20)
    p.add_argument('--warmup', type=int, default=5)
    p.add_argument('--config', default='128:3')
    p.add_argument('--trainer', default='cnn', choices=['cnn', 'iqn'])
    p.add_argument('--batch', type=int, default=64, help='per-GPU batch')
    p.add_argument('--eager', action='store_true', help='do not replay HIP graphs')
    p.add_argument('--no-cpu-baseline', action='store_true')
    p.add_argument('--no-kernel-timing', action='store_true')
    p.add_argument('--cpu-batch', type=int, default=None, help='batch of the CPU-baseline sample (default: --batch)')
    p.add_argument('--sync-bn', action='store_true', help='N>1: BatchNorm statistics of the GLOBAL batch (SyncBN)')
    p.add_argument('--overlap', choices=['off', 'on', 'auto', 'tune'], default='auto',
                   help='N>1: schedule of the two gradient-bucket all-reduces.  off (= auto, the default): serial, on the compute stream '
                        '-- with RCCL captured into the step graphs (measured on one rank: 9.50 ms/step against 9.89 with the side '
                        'stream, 9.43 without collectives); on: side stream, joined where consumed; tune: time both, keep the faster')
    p.add_argument('--backend', default='nccl', help='torch.distributed backend for N>1 (nccl = RCCL; gloo only to rehearse the DP code path)')
    p.add_argument('--share-gpu', action='store_true', help='rehearsal only: every rank uses cuda:0')
    p.add_argument('--no-graph-collectives', action='store_true',
                   help='N>1: keep the bucket all-reduces as eager calls between graph replays (the fallback of the self-launcher)')
    p.add_argument('--rehearse-rccl', action='store_true',
                   help='N=1 only: a ONE-rank RCCL group with DataParallel(rehearse=True) -- every collective of an N-rank step is '
                        'issued (bucket all-reduces on the side stream, SyncBN sums inside the graphs) and is the identity; '
                        'measures what the collectives cost the step apart from the wire')
    return p.parse_args()


def make_trainer(config, kind, batch, device):
    from tartangan_amd.models.pluggan import GAN_CONFIGS
    from tartangan_amd.trainers.cnn import CNNTrainer
    from tartangan_amd.trainers.iqn import IQNTrainer
    name, _, att = config.partition(':')
    cfg = GAN_CONFIGS[name]
    if att:
        cfg = cfg._replace(attention=(int(att),))
    cls = {'cnn': CNNTrainer, 'iqn': IQNTrainer}[kind]
    tr = cls(cls.default_args(config=cfg, batch_size=batch, device=device))
    torch.manual_seed(1234)
    tr.build_models()
    return tr, cfg


class KernelTimer:
    """HIP-event timing of every C-ABI launch of one eager step, on the launch stream."""

    def __init__(self, backend):
        self.K = backend
        self.records = []
        self._saved = {}

    def __enter__(self):
        for name in ('conv2d_fwd', 'conv2d_fwd_up2res', 'conv2d_dgrad', 'conv2d_wgrad', 'bn_train_stats', 'bn_act_fwd', 'bn_act_bwd',
                     'bn_act_dbwd', 'gemm', 'softmax_fwd', 'softmax_bwd', 'softmax_dbwd', 'up2x', 'pool2',
                     'bilinear_half_fwd', 'bilinear_half_bwd', 'add', 'channel_sum', 'adam_step', 'ema', 'attn_fwd', 'attn_bwd', 'bn_train_fwd',
                     'maxpool2_fwd', 'maxpool2_bwd', 'scale_add_dev', 'dot', 'scale_dev', 'mul', 'lrelu_bwd', 'tanh_fwd', 'tanh_bwd',
                     'row_sum', 'row_bcast', 'bce_logits', 'sumsq', 'fill', 'scale', 'channel_bcast', 'conv2d_wgrad_partials',
                     'conv2d_wgrad_reduce_batch', 'upconv3x3_fwd', 'upconv3x3_weights', 'upconv3x3_dgrad', 'upconv3x3_wgrad',
                     'upconv3x3_weights_t', 'poolconv3x3_fwd', 'poolconv3x3_dgrad', 'poolconv3x3_wgrad', 'poolconv3x3_weights',
                     'bn_train_fwd_groups', 'bn_act_bwd_groups', 'copy_channels', 'conv1x1_multi_fwd', 'conv1x1_multi_dgrad',
                     'conv1x1_multi_wgrad', 'add4', 'maxpool2_gather', 'repeat_rows', 'sum_reps', 'iqn_loss', 'iqn_cos_embed'):
            fn = getattr(self.K, name)
            self._saved[name] = fn
            setattr(self.K, name, self._wrap(name, fn))
        return self

    def _wrap(self, name, fn):
        def timed(*args):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            rc = fn(*args)
            b.record()
            self.records.append((name, args, a, b))
            return rc
        return timed

    def __exit__(self, *exc):
        for name, fn in self._saved.items():
            setattr(self.K, name, fn)

    @staticmethod
    def _empty_pair_ms(n=256):
        """Elapsed time between two events recorded back to back with nothing in between: the fixed cost that an
        event pair adds around a kernel; subtracted from every record (tiny kernels are otherwise over-counted)."""
        pairs = []
        for _ in range(n):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(); b.record()
            pairs.append((a, b))
        torch.cuda.synchronize()
        vals = sorted(a.elapsed_time(b) for a, b in pairs)
        return vals[len(vals) // 2]

    def summary(self, steps=1):
        """Per-STEP totals.  `steps` identical steps were recorded back to back: launch i of every step is the same kernel
        on the same shapes, and its time is the MEDIAN over the steps -- in an eager pass the GPU sometimes runs ahead of
        the host, and the wait for the next launch packet then lands between the two events of that launch."""
        torch.cuda.synchronize()
        base = self._empty_pair_ms()
        per = len(self.records) // steps
        same = steps > 1 and per * steps == len(self.records) and all(
            self.records[i][0] == self.records[i + k * per][0] for k in range(1, steps) for i in range(per))
        times = [max(a.elapsed_time(b) - base, 0.0) for _, _, a, b in self.records]
        if same:
            times = [sorted(times[i + k * per] for k in range(steps))[steps // 2] for i in range(per)]
            records = self.records[:per]
        else:
            times = [t / steps for t in times]
            records = self.records
        agg = {}
        for (name, args, _, _), ms in zip(records, times):
            flops = 0.0
            nbytes = 0.0
            if name == 'conv2d_fwd_up2res':       # a 3x3 forward conv whose residual is read at half the resolution
                B, Cin, Cout, H, W = args[5:10]
                flops = 2.0 * B * Cin * Cout * H * W * 9
                nbytes = 4.0 * (B * Cin * H * W + B * Cout * H * W + Cin * Cout * 9) + 1.0 * B * Cout * H * W
                name = 'conv2d_fwd'
            elif name in CONV_DIMS:
                B, Cin, Cout, H, W, ks = args[CONV_DIMS[name]]
                flops = 2.0 * B * Cin * Cout * H * W * ks * ks
                # every operand once: input, output (or the two activations of wgrad) and the filter
                nbytes = 4.0 * (B * Cin * H * W + B * Cout * H * W + Cin * Cout * ks * ks)
                if name == 'conv2d_fwd' and args[3] is not None:
                    nbytes += 4.0 * B * Cout * H * W          # fused residual read
            STRIDE2 = {'upconv3x3_fwd': 5, 'poolconv3x3_fwd': 5, 'upconv3x3_dgrad': 3, 'poolconv3x3_dgrad': 3,
                       'upconv3x3_wgrad': 5, 'poolconv3x3_wgrad': 5}
            if name in STRIDE2:                   # (..., B, Cin, Cout, H, W, ...) with H x W the low-resolution plane: 16 taps
                B, Cin, Cout, H, W = args[STRIDE2[name]:STRIDE2[name] + 5]
                flops = 2.0 * B * Cin * Cout * H * W * 16
                hi = Cin if name.startswith('poolconv') else Cout      # channels of the high-resolution (2H x 2W) tensor
                nbytes = 4.0 * (B * hi * 4 * H * W + B * (Cin + Cout - hi) * H * W + Cin * Cout * 16)
            if name in ('conv2d_fwd', 'conv2d_dgrad', 'conv2d_wgrad_partials', 'conv2d_wgrad') and len(args) >= CONV_DIMS[name].stop \
                    and args[CONV_DIMS[name]][5] == 1:
                name = name + '_1x1'         # bandwidth-bound (AI = Cin Cout / (2 (Cin + Cout)) FLOP/B << the MFMA ridge)
            d = agg.setdefault(name, dict(ms=0.0, launches=0, flops=0.0, bytes=0.0))
            d['ms'] += ms
            d['launches'] += 1
            d['flops'] += flops
            d['bytes'] += nbytes
        if not same and steps > 1:                # steps differed in their launch sequence: plain per-step averages
            for d in agg.values():
                d['launches'] //= steps
                d['flops'] /= steps
                d['bytes'] /= steps
        return agg


# where (B, Cin, Cout, H, W, ks) sit in each conv entry point's argument list (include/tartangan_amd.h)
CONV_DIMS = {'conv2d_fwd': slice(5, 11), 'conv2d_dgrad': slice(3, 9), 'conv2d_wgrad': slice(6, 12),
             'conv2d_wgrad_partials': slice(4, 10)}


TRAFFIC_FILE = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'profiles', 'r03_hbm_traffic.json')


def hbm_traffic_per_launch(a):
    """HBM bytes per launch of the conv fwd/dgrad family from the committed PMC summary (rocprofv3 --pmc FETCH_SIZE /
    WRITE_SIZE passes over this same command, tools/pmc_traffic.sh, corrected as the MI355X guide prescribes).  Counters
    cannot be collected from inside the timed process, so the figure comes from that file -- and only if the file was
    measured on the kernel source this process runs (sha256 of csrc/conv.hip + csrc/wino.h stamped into it) and on this workload;
    otherwise null, with the reason beside it.  -> (bytes or None, provenance dict)"""
    import hashlib
    import subprocess
    if not os.path.exists(TRAFFIC_FILE):
        return None, {'traffic_source': None}
    with open(TRAFFIC_FILE) as f:
        k = json.load(f)
    here = os.path.dirname(os.path.abspath(__file__))
    sha = hashlib.sha256(b''.join(open(os.path.join(here, 'tartangan_amd', 'csrc', f), 'rb').read() for f in ('conv.hip', 'wino.h'))).hexdigest()
    try:
        blob = subprocess.run(['git', 'hash-object', TRAFFIC_FILE], capture_output=True, text=True, cwd=here).stdout.strip()
    except Exception:
        blob = None
    prov = {'traffic_source': 'profiles/r03_hbm_traffic.json', 'traffic_file_git_blob': blob or None,
            'traffic_measured_on_conv_hip_sha256': k.get('conv_hip_sha256', '')[:16], 'conv_hip_sha256': sha[:16]}
    if k.get('conv_hip_sha256') != sha:
        prov['traffic_note'] = 'stale: csrc/conv.hip / wino.h changed since the counters were collected'
        return None, prov
    if not (a.config == '128:3' and a.trainer == 'cnn' and a.batch == 64):
        prov['traffic_note'] = 'measured on the default workload only'
        return None, prov
    return k['conv_family_fwd_dgrad']['hbm_bytes_per_launch'], prov


# the MFMA convolution kernels that produce activations / activation gradients: C-ABI entry point -> kernel name in the
# PMC summary.  (The stride-2 kernels serve the generator's up-convs and the discriminator's pooled convs.)
CONV_FAMILY = {'conv2d_fwd': 'conv_wino_dma_kernel / conv_dma_kernel / conv_fwd_kernel (fwd)',
               'conv2d_dgrad': 'conv_wino_dma_kernel / conv_dma_kernel / conv_fwd_kernel (dgrad)',
               'upconv3x3_fwd': 'conv_upfwd_dma_kernel', 'poolconv3x3_dgrad': 'conv_upfwd_dma_kernel',
               'upconv3x3_dgrad': 'conv_poolwino_dma_kernel / conv_upT_dma_kernel', 'poolconv3x3_fwd': 'conv_poolwino_dma_kernel / conv_upT_dma_kernel'}


def _cpu_sample(config, kind, batch, threads, n):
    from oracle import sagan_cpu as O
    name, _, att = config.partition(':')
    prev = torch.get_num_threads()
    torch.set_num_threads(threads)
    try:
        torch.manual_seed(1234)
        tr = O.OracleTrainer(name, kind, batch, attention=(int(att),) if att else None)
        size = tr.cfg.base_size * 2 ** len(tr.cfg.blocks)
        imgs = torch.rand(batch, 3, size, size) * 2 - 1
        tr.train_batch(imgs)                       # untimed first step (allocator / thread-pool warm-up)
        t0 = time.perf_counter()
        for _ in range(n):
            tr.train_batch(imgs)
        return batch * n / (time.perf_counter() - t0)
    finally:
        torch.set_num_threads(prev)


def physical_cores():
    """(physical cores, logical CPUs this process may run on)."""
    logical = len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else (os.cpu_count() or 1)
    cores = set()
    try:
        phys = core = None
        for line in open('/proc/cpuinfo'):
            if line.startswith('physical id'):
                phys = line.split(':')[1].strip()
            elif line.startswith('core id'):
                core = line.split(':')[1].strip()
            elif not line.strip():
                if phys is not None and core is not None:
                    cores.add((phys, core))
                phys = core = None
    except OSError:
        pass
    return (min(len(cores), logical) if cores else logical), logical


def cpu_baseline(config, kind, batch):
    """The CPU oracle (kind "port": oracle/sagan_cpu.py, plain PyTorch CPU fp32, pinned to the reference by the golden
    fixtures) timed on this host at the GPU line's own workload and batch.  `value` is the BEST of a short sweep over
    thread counts {8, 16, 32, 64, physical cores} (each: 1 untimed + 1 timed step; more threads than the oracle's small
    layers can use only add contention -- round 2's all-threads sample lost to its own 8-thread run), `cores` the thread
    count that achieved it; the whole sweep is ~20-40 s of host time."""
    phys, logical = physical_cores()
    counts = sorted({c for c in (8, 16, 32, 64, phys) if c <= max(phys, 8) and c <= max(logical, 8)})
    sweep, worse = {}, 0
    for c in counts:
        v = _cpu_sample(config, kind, batch, c, 1)
        if sweep and v < max(sweep.values()):
            worse += 1
        sweep[c] = round(v, 3)
        if worse >= 2:                   # past the knee: larger counts only cost bench time
            break
    best = max(sweep, key=sweep.get)
    return dict(value=sweep[best], unit='images/s', cores=best, kind='port', physical_cores=phys, logical_cpus=logical,
                thread_sweep={str(k): v for k, v in sweep.items()},
                sample=f'best of a thread sweep {list(sweep)}: 1 timed step (+1 untimed) each of the {config} {kind} step at batch '
                       f'{batch}, oracle/sagan_cpu.py (plain PyTorch CPU fp32); host has {phys} physical cores / {logical} logical CPUs')


def _free_port():
    import socket
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


def self_launch(a):
    """`python bench.py --gpus N` from a bare shell (no WORLD_SIZE): start the N ranks as FRESH child processes through
    torch.distributed.run and relay rank 0's JSON line.  This parent never touches the GPU (no torch.cuda call, no HIP
    library load), so nothing GPU-initialised is ever re-exec'ed or forked.  Exit code = the children's."""
    import subprocess

    def run(extra):
        cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', f'--nproc-per-node={a.gpus}',
               '--master-addr', '127.0.0.1', '--master-port', str(_free_port()), os.path.abspath(__file__), *sys.argv[1:], *extra]
        env = dict(os.environ)
        env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')        # dmabuf IPC: what RCCL needs on this host driver
        env.setdefault('OMP_NUM_THREADS', str(max(1, (os.cpu_count() or 8) // a.gpus)))
        proc = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, text=True)
        lines = [ln for ln in proc.stdout.splitlines() if ln.startswith('{"metric"')]
        for ln in proc.stdout.splitlines():
            if not ln.startswith('{"metric"'):
                print(ln, file=sys.stderr)
        return proc.returncode, lines

    rc, lines = run([])
    if (rc != 0 or not lines) and not (a.overlap == 'off' and a.no_graph_collectives):
        # one fallback, for a collective library that fails FAST with collectives inside graphs or on the side stream: the
        # serial schedule with eager collectives between graph replays (a run that hangs is not retried -- the driver's limit ends it)
        print(f'bench.py: ranks exited with {rc}; retrying once with --overlap off --no-graph-collectives', file=sys.stderr)
        rc, lines = run(['--overlap', 'off', '--no-graph-collectives'])
    if rc == 0 and len(lines) == 1:
        print(lines[0], flush=True)
        return 0
    print(f'bench.py: ranks exited with {rc}, {len(lines)} result line(s)', file=sys.stderr)
    return rc or 1


def main():
    a = parse()
    if a.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        sys.exit(self_launch(a))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if world != a.gpus:
        sys.exit(f'bench.py: --gpus {a.gpus} but WORLD_SIZE={world}; launch one rank per GPU '
                 f'(python bench.py --gpus N starts them itself)')
    if a.share_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    rehearse = a.rehearse_rccl and world == 1
    if rehearse:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', str(_free_port()))
        os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        dist.init_process_group('nccl', rank=0, world_size=1, device_id=torch.device('cuda', local_rank))
    if world > 1:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        if a.backend == 'nccl':
            dist.init_process_group('nccl', device_id=torch.device('cuda', local_rank))
        else:
            dist.init_process_group(a.backend)

    from tartangan_amd import backend
    K = backend.get()                          # fails loudly without the HIP library
    tr, cfg = make_trainer(a.config, a.trainer, a.batch, 'cuda')
    if world > 1 or rehearse:
        from tartangan_amd.parallel import DataParallel
        if not a.eager:
            tr.enable_graphs()
        overlap = a.overlap == 'on'
        dp = DataParallel(tr, sync_bn=a.sync_bn, overlap=overlap, rehearse=rehearse, graph_buckets=not a.no_graph_collectives)
    elif not a.eager:
        tr.enable_graphs()
    size = tr.g.max_size
    g = torch.Generator().manual_seed(1234)
    imgs_global = torch.rand(a.batch * world, 3, size, size, generator=g) * 2 - 1
    imgs = imgs_global[rank * a.batch:(rank + 1) * a.batch].cuda()     # resident in HBM before timing
    torch.manual_seed(1234)

    warm = max(a.warmup, 0 if a.eager else 2)  # graph mode: step 1 eager (records RNG plan), step 2 captures
    for _ in range(warm):
        logs = tr.train_batch(imgs)
    if world > 1 and a.overlap == 'tune':       # (untimed) keep the side-stream all-reduce only if it is not slower here
        dp.autotune_overlap(lambda: tr.train_batch(imgs))

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    fence()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        logs = tr.train_batch(imgs)
    fence()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device='cuda')
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t)
    ms_per_step = dt / a.steps * 1e3
    value = a.batch * world * a.steps / dt

    agg = {}
    if not a.no_kernel_timing:
        # instrumented eager pass of the same step (same kernels, same shapes) with HIP events on the launch stream;
        # every rank runs it (the step's collectives must stay matched), rank 0 reports its own kernels
        saved = (getattr(tr, '_graphs', None), getattr(tr, '_graph_requested', False), tr.rng_feed.mode)
        tr._graphs, tr._graph_requested = None, False
        tr.rng_feed.mode = 'off'
        timed_steps = 3                      # a few steps: one step's 196 conv launches are a noisy sample
        with KernelTimer(K) as kt:
            for _ in range(timed_steps):
                tr.train_batch(imgs)
        agg = kt.summary(timed_steps)
        tr._graphs, tr._graph_requested, tr.rng_feed.mode = saved

    if rank == 0:
        out = {
            'metric': 'images/sec (G+D step)', 'value': round(value, 2), 'unit': 'images/s', 'n_gpus': world,
            'steps': a.steps, 'warmup': warm, 'ms_per_step': round(ms_per_step, 4), 'higher_is_better': True,
            'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
            'config': {'workload': f'{a.config} SA-GAN {a.trainer} G+D step (R1 penalty, Adam x2, EMA), '
                                   f'{size}x{size} RGB, batch {a.batch}/GPU, global batch {a.batch * world}',
                       'trainer': a.trainer, 'gan_config': a.config, 'global_batch': a.batch * world,
                       'parallelism': f'dp{world}' + ('' if world == 1 and not rehearse else
                                                      f' ({"global-batch (synchronised)" if a.sync_bn else "local-batch"} BatchNorm, '
                                                      f'flat-bucket {dp.collective_name} all-reduce x2'
                                                      f'{" on a side stream" if dp.overlap else (", captured into the step graphs" if dp.buckets_in_graph and not a.eager else "")})'),
                       'hip_graphs': not a.eager,
                       **({'rehearsal': 'one-rank RCCL group, every collective issued'} if rehearse else {}),
                       **({} if world == 1 and not rehearse else {'rccl_ranks': dist.get_world_size(), 'collective_backend': dist.get_backend(),
                                                 'sync_bn': bool(a.sync_bn), 'overlap': bool(dp.overlap)})},
            'final_losses': {k: round(v, 6) for k, v in logs.items()},
        }
        flop_img = FLOP_PER_IMAGE.get(a.config)
        if flop_img and a.trainer == 'cnn':
            tf = flop_img * value / world / 1e12
            # MODEL FLOPs (the reference formulation's) per second -- the step-level utilisation figure; the
            # engine executes fewer (16-tap stride-2 forms, composed from-RGB): see executed_conv_tflops below
            out['roofline_step'] = {'bound': 'mfma', 'achieved': round(tf, 3), 'peak': MFMA_F32_PEAK_TFLOPS,
                                    'unit': 'TFLOP/s', 'frac': round(tf / MFMA_F32_PEAK_TFLOPS, 4),
                                    'flop_per_image': flop_img, 'flops': 'model (reference formulation)'}
        if agg:
            conv = {k: sum(agg[n][k] for n in CONV_FAMILY if n in agg) for k in ('ms', 'launches', 'flops', 'bytes')}
            traffic, traffic_prov = hbm_traffic_per_launch(a)
            ach = conv['flops'] / (conv['ms'] * 1e-3) / 1e12
            out['roofline'] = {'bound': 'mfma', 'kernel': 'MFMA 3x3 / stride-2 conv family (activations and activation gradients): '
                                                          'conv_wino_dma_kernel (Winograd F(2x2,3x3), fwd + dgrad), conv_dma_kernel, '
                                                          'conv_upfwd_dma_kernel, conv_poolwino_dma_kernel (9-frequency pooled form), conv_upT_dma_kernel',
                               'flops': 'algorithmic (direct-form multiply-adds of the layer); the Winograd kernels issue 2.25x / 1.78x fewer',
                               'achieved': round(ach, 3), 'peak': MFMA_F32_PEAK_TFLOPS, 'unit': 'TFLOP/s',
                               'frac': round(ach / MFMA_F32_PEAK_TFLOPS, 4),
                               'traffic': traffic, **traffic_prov,
                               'algorithmic_bytes_per_launch': round(conv['bytes'] / conv['launches']),
                               'launches_per_step': conv['launches'],
                               'avg_launch_ms': round(conv['ms'] / conv['launches'], 5),
                               'algorithmic_flop_per_step': conv['flops']}
            one = {k: sum(agg[n][k] for n in ('conv2d_fwd_1x1', 'conv2d_dgrad_1x1') if n in agg) for k in ('ms', 'launches', 'bytes')}
            if one['launches']:          # the 1x1 convolutions are priced against the HBM roof
                gbs = one['bytes'] / (one['ms'] * 1e-3) / 1e9
                out['roofline_1x1'] = {'bound': 'hbm', 'kernel': '1x1 convolutions (fwd + dgrad): conv_fwd_kernel<.., 1, ..>, conv1x1_direct_kernel',
                                       'achieved': round(gbs, 1), 'peak': HBM_PEAK_GBS, 'unit': 'GB/s', 'frac': round(gbs / HBM_PEAK_GBS, 4),
                                       'launches_per_step': one['launches'], 'avg_launch_ms': round(one['ms'] / one['launches'], 5)}
            total = sum(d['ms'] for d in agg.values())
            out['kernel_time_ms'] = {k: {'ms': round(d['ms'], 4), 'launches': d['launches'],
                                         **({'tflops': round(d['flops'] / (d['ms'] * 1e-3) / 1e12, 2)} if d['flops'] else {})}
                                     for k, d in sorted(agg.items(), key=lambda kv: -kv[1]['ms'])}
            out['kernel_time_ms']['_sum_of_timed_launches'] = round(total, 4)
        if agg and 'roofline_step' in out:
            ex = sum(d['flops'] for d in agg.values())
            out['roofline_step']['executed_flop_per_step'] = ex
            out['roofline_step']['executed_tflops'] = round(ex / (ms_per_step * 1e-3) / 1e12, 3)
        if not a.no_cpu_baseline and world == 1:       # reported at N = 1 only
            out['cpu_baseline'] = cpu_baseline(a.config, a.trainer, a.cpu_batch or a.batch)
        print(json.dumps(out), flush=True)
    if world > 1 or rehearse:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
