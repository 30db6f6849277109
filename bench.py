#!/usr/bin/env python3
"""bench.py — (photo, render) pairs/s of the 3D-FM GAN forward hot path on MI355X, with the upfirdn2d roofline.

    python bench.py --gpus N --steps K --warmup W

N>1 under torch.distributed.run (RANK / WORLD_SIZE in the environment): this process is one of the N ranks.  N>1 started
plainly: this process touches no GPU, starts the N ranks itself as children (one per GPU, rendezvous on 127.0.0.1) and
relays rank 0's line — the reference goes multi-GPU from one command too (train_3_encoder.py:355-362).  A rank count that
differs from --gpus is an error, never a silently smaller run.

A "step" is one forward of the hot path over one batch of synthetic (photo, render) pairs per GPU:
E_Tsr + E_W + E_W_Plus on 256^2 images -> co-modulation -> Generator -> image, fp32, eval-mode BatchNorm, no_grad.
Default workload `pairs1024` is BASELINE config 4's per-GPU shard (B=8/GPU, Generator(1024), 18 styles — encoders on
256^2), the configuration the headline upfirdn2d call ([B*32,1025,1025] -> [B*32,1024,1024]) lives in;
`pairs256` (B=32/GPU, Generator(256)) is timed in the same run and reported as `pairs_per_s_256`, and BASELINE
config 2 (Generator(256) alone on random W+, B=32) as `synthesis256_images_per_s`.
Forward shards over ranks by batch with no collective: weak scaling, value = all ranks' pairs / max time.
Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time
import types

ROOT = os.path.dirname(os.path.abspath(__file__))
for _p in (ROOT, os.path.join(ROOT, '3d-fm-gan_amd'), os.path.join(ROOT, 'tests')):
    if _p not in sys.path:
        sys.path.insert(0, _p)
os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)
FP32_MFMA_PEAK_TF = 157.3    # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 peak

WORKLOADS = {
    'pairs1024': dict(size=1024, batch=8, desc='cfg4 shard: E_Tsr+E_W+E_W_Plus(18 styles) on 256^2 -> co-mod -> '
                      'Generator(1024) forward, fp32, B=8/GPU'),
    'pairs256': dict(size=256, batch=32, desc='E_Tsr+E_W+E_W_Plus(14 styles) on 256^2 -> co-mod -> Generator(256) '
                     'forward, fp32, B=32/GPU'),
    # BASELINE config 3: forward + backward through the three encoders and the Generator (loss = L1 to a random target),
    # eval-mode BatchNorm; with N > 1 the gradients are all-reduced by DDP over RCCL.
    'train256': dict(size=256, batch=16, desc='cfg3: full 3-encoder forward+backward @256^2, fp32, B=16/GPU, '
                     'grad all-reduce (DDP/RCCL) when N>1'),
    # One iteration of train() (train_3_encoder.py:756-828) in fp32 with the losses available offline: D logistic loss,
    # R1 every 16, G non-saturating + L1, path length every 4 on batch/2, EMA; Adam on G+encoders and on D.
    'trainstep256': dict(size=256, batch=16, desc='train() iteration @256^2, fp32, B=16/GPU: D loss + R1/16 + G adv+L1 + '
                         'path length/4 on B/2 + EMA, Adam; DDP/RCCL all-reduce of G/encoder and D gradients when N>1'),
    # BASELINE config 5: the train() iteration at 1024^2 — Generator(1024), Discriminator(1024), 18 styles —
    # with the LPIPS (net-lin VGG16) and ArcFace (resnet_face18) loss networks in the G step at the reference's weights
    # (lpips_loss_lambda=3, face_id_loss_lambda=30, train_3_encoder_hyperparams.py:62,65).  Their pretrained weights are
    # not available offline: both are built with their own initialisation and are LOAD GENERATORS, not parity rows.
    'trainstep1024': dict(size=1024, batch=8, loss_nets=True,
                          desc='cfg5: train() iteration @1024^2, fp32, B=8/GPU: D(1024) loss + R1/16 + G adv + L1 + LPIPS('
                          'VGG16 topology, random init: load only) + ArcFace(resnet_face18 topology, random init: load only) '
                          '+ path length/4 on B/2 + EMA, Adam; DDP/RCCL all-reduce when N>1'),
    # The same iteration with reduced-precision contractions — a LABELLED EXTENSION, never the headline (the reference
    # trains in fp32; SURVEY §8c states 5e-2 on [-1,1] images for this configuration): torch.autocast(bf16) around the
    # whole iteration (every MIOpen convolution / linear of the encoders, D and the loss networks in bf16; this repo's
    # HIP ops cast their inputs back to fp32) and the modulated conv's forward / data-gradient contractions on
    # v_mfma_f32_32x32x16_bf16 (bf16 operands, fp32 accumulation; weight gradients and optimiser state stay fp32).
    'trainstep1024_bf16': dict(size=1024, batch=8, loss_nets=True, precision='bf16',
                               desc='cfg5, bf16 leg: trainstep1024 under torch.autocast(bfloat16) + bf16-operand / fp32-'
                               'accumulate modulated conv (csrc/modconv_bf16.hip); fp32 master weights and optimiser'),
}
# the lazy regularisers recur every 16 (R1) and 4 (path length) iterations: any 16 consecutive iterations hold exactly
# one R1 step and four path-length steps, so a timed window of 16 IS the amortised cost of an iteration
TRAINSTEP_WINDOW = 16


_T0 = time.time()


def log(msg):
    """Progress on stderr (rank 0): the JSON line on stdout stays alone, and a long run is never silent."""
    if int(os.environ.get('RANK', '0')) == 0:
        print(f'[bench {time.time() - _T0:6.1f}s] {msg}', file=sys.stderr, flush=True)


class LaunchTimer:
    """HIP events on the launch stream (torch's current stream IS the stream _native launches on) around the
    launches selected by `want`; times are read after the region has been synchronised."""

    wants_paths = True      # _native records which kernel served each fused blur (roofline.kernel names what ran)

    def __init__(self, want):
        self.want = want
        self.pairs = []

    def begin(self, name, info):
        if not self.want(name, info):
            return None
        s = torch.cuda.Event(enable_timing=True)
        s.record()
        return (name, info, s)

    def end(self, tok):
        if tok is None:
            return
        e = torch.cuda.Event(enable_timing=True)
        e.record()
        self.pairs.append(tok + (e,))

    def summary(self):
        out = {}
        for name, info, s, e in self.pairs:
            out.setdefault((name, info), []).append(s.elapsed_time(e))
        return {k: (sum(v) / len(v), len(v)) for k, v in out.items()}


def build_models(size, device):
    import stylegan2
    import resnet_encoder
    from psp_encoder_model.encoders import psp_encoders
    n_latent = int(round(__import__('math').log2(size))) * 2 - 2
    torch.manual_seed(0)
    nets = dict(
        e_tsr=resnet_encoder.resnet18(tensor_encoding=True, tensor_transform=False),
        e_w=resnet_encoder.resnet18(tensor_encoding=False, tensor_transform=False),
        e_wp=psp_encoders.GradualStyleEncoder(18, 'ir_se', types.SimpleNamespace(input_nc=3, n_styles=n_latent)),
        g=stylegan2.Generator(size, 512, 8),
    )
    for m in nets.values():
        m.eval().requires_grad_(False)
    return {k: m.to(device) for k, m in nets.items()}


def make_step(nets, batch, device, rank):
    from Util.network_util import Forward_Inference_3_Encoder
    gen = torch.Generator(device='cpu').manual_seed(1234 + rank)
    photo = (torch.rand(batch, 3, 256, 256, generator=gen) * 2 - 1).to(device)     # resident in HBM before timing
    render = (torch.rand(batch, 3, 256, 256, generator=gen) * 2 - 1).to(device)

    def forward(p, r):
        with torch.no_grad():
            return Forward_Inference_3_Encoder(p, r, nets['e_tsr'], nets['e_w'], nets['e_wp'], nets['g'])

    def step():
        return forward(photo, render)

    step.forward = forward
    return step, (photo, render)


def make_train_step(nets, batch, device, rank, world):
    """forward + backward of the (photo, render) -> image path; gradients w.r.t. every encoder and generator weight."""
    from Miscellaneous import distributed as D
    from Util.network_util import Forward_Inference_3_Encoder
    for m in nets.values():
        m.requires_grad_(True)
    # only the Generator has parameters without a gradient (mapping network, constant input)
    wrapped = {k: D.data_parallel(m, device, find_unused_parameters=(k == 'g')) for k, m in nets.items()}
    gen = torch.Generator(device='cpu').manual_seed(1234 + rank)
    photo = (torch.rand(batch, 3, 256, 256, generator=gen) * 2 - 1).to(device)
    render = (torch.rand(batch, 3, 256, 256, generator=gen) * 2 - 1).to(device)
    target = (torch.rand(batch, 3, nets['g'].size, nets['g'].size, generator=gen) * 2 - 1).to(device)
    params = [p for m in nets.values() for p in m.parameters()]

    def step():
        for p in params:
            p.grad = None
        img = Forward_Inference_3_Encoder(photo, render, wrapped['e_tsr'], wrapped['e_w'], wrapped['e_wp'], wrapped['g'])
        (img - target).abs().mean().backward()

    return step, (photo, render)


def make_trainstep(nets, batch, device, rank, world, size, loss_nets=False, precision='f32'):
    """A full training iteration (3d-fm-gan_amd/train_3_encoder.py::Trainer.step) on synthetic pairs."""
    import stylegan2
    import train_3_encoder as T
    torch.manual_seed(1)
    d = stylegan2.Discriminator(size).to(device)
    lp, fr = T.Module_Fix_Setup(T.default_args(), device) if loss_nets else (None, None)
    gen = torch.Generator(device='cpu').manual_seed(1234 + rank)
    photo = (torch.rand(batch, 3, 256, 256, generator=gen) * 2 - 1).to(device)
    render = (torch.rand(batch, 3, 256, 256, generator=gen) * 2 - 1).to(device)
    ref = (torch.rand(batch, 3, size, size, generator=gen) * 2 - 1).to(device)
    tr = T.Trainer(dict(G=nets['g'], E_Tsr=nets['e_tsr'], E_W=nets['e_w'], E_W_Plus=nets['e_wp'], D=d),
                   T.default_args(rec_batch=batch), device, lpips_model=lp, face_rec_model=fr)

    if precision == 'bf16':
        from op import _native

        def step():
            with torch.autocast('cuda', dtype=torch.bfloat16), _native.modconv_precision('bf16'):
                tr.step(photo, render, ref)
    else:
        def step():
            tr.step(photo, render, ref)

    return step, tr


def warm_miopen_cache():
    """Point MIOpen at the kernel cache shipped next to libfmgan_hip.so, when there is one (a build artefact like the
    .so: git-ignored, made by tools/warm_miopen_cache.sh on an MI355X box with this image).  It only saves the
    minutes of hipRTC compilation MIOpen otherwise spends on first use of every convolution; timed regions start after
    warm-up either way.  Must run before the first convolution."""
    d = os.path.join(ROOT, '3d-fm-gan_amd', 'miopen_cache')
    if os.path.isdir(os.path.join(d, 'db')):
        os.environ.setdefault('MIOPEN_USER_DB_PATH', os.path.join(d, 'db'))
        os.environ.setdefault('MIOPEN_CUSTOM_CACHE_DIR', os.path.join(d, 'cache'))
        return True
    return False


def train_leg(name, steps, world, timeout_s):
    """Run `bench.py --workload <name>` as a child (same ranks, next rendezvous port) and return its JSON record, or
    {'skipped': reason}.  Every rank calls this; rank 0's child prints the line."""
    import subprocess
    env = dict(os.environ)
    if world > 1:
        env['MASTER_PORT'] = str(int(env.get('MASTER_PORT', '29500')) + {'train256': 7, 'trainstep256': 13, 'trainstep1024': 19}.get(name, 23))
    cmd = [sys.executable, os.path.abspath(__file__), '--gpus', str(world), '--workload', name, '--steps', str(steps),
           '--warmup', '2', '--no-cpu-baseline']
    log(f'{name}: child process, time box {timeout_s:.0f} s')
    t0 = time.time()
    try:
        pr = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=None, text=True, timeout=timeout_s)
    except subprocess.TimeoutExpired:
        return {'skipped': f'{name}: not finished within {timeout_s:.0f} s (MIOpen compiling backward kernels on a cold cache)'}
    lines = [ln for ln in pr.stdout.splitlines() if ln.startswith('{')]
    if pr.returncode != 0 or not lines:
        return {'skipped': f'{name}: child exited with code {pr.returncode}' if int(os.environ.get('RANK', '0')) == 0 else 'n/a'}
    log(f'{name}: done in {time.time() - t0:.0f} s')
    return json.loads(lines[-1])


def dist_info(world):
    """What the process group itself reports (not what was asked for): backend and rank count."""
    if world > 1 and dist.is_initialized():
        return {'backend': 'rccl (torch backend "nccl")' if dist.get_backend() == 'nccl' else dist.get_backend(),
                'ranks': dist.get_world_size(), 'launcher': 'bench.py self-launch' if os.environ.get('FMGAN_SELF_LAUNCHED')
                else 'external (torch.distributed.run)'}
    return {'backend': None, 'ranks': 1, 'launcher': None}


def timed(step, steps, warmup, world):
    for _ in range(warmup):
        step()
    gpu = torch.cuda.is_available()
    if world > 1:
        dist.barrier()
    if gpu:
        torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    if gpu:
        torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        dev = 'cuda' if dist.get_backend() == 'nccl' else 'cpu'
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = t.item()
    return dt


def committed_traffic():
    import glob
    import hashlib
    for tf in sorted(glob.glob(os.path.join(ROOT, 'profiles', 'r*_headline_traffic.json')), reverse=True):
        rec = json.load(open(tf))
        src = rec.get('kernel_source')
        if not src or not os.path.exists(os.path.join(ROOT, src)):
            continue
        sha = hashlib.sha256(open(os.path.join(ROOT, src), 'rb').read()).hexdigest()
        if sha != rec.get('kernel_source_sha256'):
            return None, f'{os.path.relpath(tf, ROOT)} is stale: {src} changed since the counters were collected'
        return rec['hbm_bytes_per_launch'], f"{os.path.relpath(tf, ROOT)} ({rec['source']}; kernel {rec.get('kernel')})"
    return None, None


def host_cores():
    """Cores this process may actually use: cgroup quota if set, else the affinity mask, capped at the 16-core
    share a one-GPU box gets (FMGAN_CPU_CORES overrides)."""
    if os.environ.get('FMGAN_CPU_CORES'):
        return int(os.environ['FMGAN_CPU_CORES'])
    n = os.cpu_count() or 1
    try:
        n = len(os.sched_getaffinity(0))
    except Exception:
        pass
    try:
        quota, period = open('/sys/fs/cgroup/cpu.max').read().split()
        if quota != 'max':
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        pass
    return min(n, 16)


def cpu_baseline(nets, inputs, size, budget_s=20.0):
    """The reference's pure-PyTorch CPU path as restated in oracle/torch_oracle.py (pinned to the reference by
    tests/test_oracle_golden.py), on this box's host cores, B=1 pairs of the same workload until ~budget_s."""
    from oracle import torch_oracle as T
    cores = host_cores()
    torch.set_num_threads(cores)
    sds = {k: {n: v.detach().cpu() for n, v in m.state_dict().items()} for k, m in nets.items()}
    photo, render = (t[:1].cpu() for t in inputs)
    n, t0 = 0, time.perf_counter()
    with torch.no_grad():
        while True:
            T.forward_inference_3_encoder(photo, render, sds['e_tsr'], sds['e_w'], sds['e_wp'], sds['g'], size)
            n += 1
            if time.perf_counter() - t0 > budget_s or n >= 16:
                break
    dt = time.perf_counter() - t0
    out = dict(value=n / dt, unit='pairs/s', cores=cores, cores_present=os.cpu_count(),
               cores_note='threads used = the CPU share of a one-GPU box (cgroup quota / affinity, capped at 16); the '
                          'socket has more cores than this process may use', kind='port', cpu_model=cpu_model(),
               sample=f'{n} pairs at B=1 through oracle/torch_oracle.py (reference CPU path restated: F.conv2d '
                      f'modconv + upfirdn2d_native + CPU fused_leaky_relu), Generator({size}), {dt:.1f} s')
    out['items'] = cpu_baseline_items(sds, photo, render, size)
    return out


def cpu_model():
    try:
        for line in open('/proc/cpuinfo'):
            if line.startswith('model name'):
                return line.split(':', 1)[1].strip()
    except Exception:
        pass
    return 'unknown'


def _rate(fn, units, budget_s, max_n):
    fn()                                   # first call pays allocator / thread-pool start-up
    n, t0 = 0, time.perf_counter()
    while True:
        fn()
        n += 1
        if time.perf_counter() - t0 > budget_s or n >= max_n:
            break
    return units * n / (time.perf_counter() - t0), n


def cpu_baseline_items(sds, photo, render, size):
    """The other CPU-fallback baselines BASELINE.md §3 lists, each on a bounded sample (same oracle, same cores):
    cfg1 pair @256^2, synthesis-only @256^2 / @1024^2, and the two ops' algorithmic GB/s at the headline shapes
    (upfirdn2d_native, op/upfirdn2d.py:168-209; CPU fused_leaky_relu, op/fused_act.py:114-125)."""
    import math
    from oracle import torch_oracle as T
    items = {}
    log('cpu baseline items')
    with torch.no_grad():
        # ops at the headline shapes, B=1
        x = torch.randn(1, 32, 1025, 1025)
        k = T.make_kernel([1, 3, 3, 1]) * 4
        r, n = _rate(lambda: T.upfirdn2d(x, k, pad=(1, 1)), 4.0 * 32 * (1025 * 1025 + 1024 * 1024) / 1e9, 4.0, 8)
        items['upfirdn2d_native_blur_1x32x1025x1025'] = dict(value=r, unit='GB/s (algorithmic in+out)', calls=n)
        x3 = torch.randn(1, 3, 512, 512)
        r, n = _rate(lambda: T.upfirdn2d(x3, k, up=2, pad=(2, 1)), 4.0 * 3 * (512 * 512 + 1024 * 1024) / 1e9, 2.0, 16)
        items['upfirdn2d_native_up2_1x3x512x512'] = dict(value=r, unit='GB/s (algorithmic in+out)', calls=n)
        xa, ba = torch.randn(1, 32, 1024, 1024), torch.randn(32)
        r, n = _rate(lambda: T.fused_leaky_relu(xa, ba), 8.0 * xa.numel() / 1e9, 3.0, 32)
        items['fused_leaky_relu_1x32x1024x1024'] = dict(value=r, unit='GB/s (algorithmic read+write)', calls=n)
        del x, xa
        # synthesis only (cfg2's network at B=1) and the cfg1 pair
        for sz in (256, 1024):
            if sz == size:
                sd_g = sds['g']
            else:
                import stylegan2
                torch.manual_seed(0)
                sd_g = {k_: v.detach() for k_, v in stylegan2.Generator(sz, 512, 8).state_dict().items()}
            n_latent = int(round(math.log2(sz))) * 2 - 2
            lat, tsr = torch.randn(1, n_latent, 512), torch.randn(1, 512, 4, 4)
            r, n = _rate(lambda: T.generator_forward(sd_g, sz, lat, external_input_tensor=tsr), 1.0, 5.0, 8)
            items[f'synthesis_only_{sz}_B1'] = dict(value=r, unit='images/s', calls=n)
            if sz == 256:
                sd_wp = sds['e_wp']
                if size != 256:      # 14-style pSp encoder for the 256^2 generator
                    from psp_encoder_model.encoders import psp_encoders
                    torch.manual_seed(0)
                    sd_wp = {k_: v.detach() for k_, v in psp_encoders.GradualStyleEncoder(
                        18, 'ir_se', types.SimpleNamespace(input_nc=3, n_styles=14)).state_dict().items()}
                r, n = _rate(lambda: T.forward_inference_3_encoder(photo, render, sds['e_tsr'], sds['e_w'], sd_wp, sd_g, 256),
                             1.0, 6.0, 8)
                items['cfg1_pair_256_B1'] = dict(value=r, unit='pairs/s', calls=n)
    return items


def self_launch(args, argv):
    """`python bench.py --gpus N` without a launcher: start the N ranks as children of a parent that never initialises
    the GPU (no exec from a process that did), one rank per GPU, and exit with the worst child's code.  Rank 0's stdout
    (the JSON line) is the parent's stdout; every rank's stderr passes through."""
    import socket
    import subprocess
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), LOCAL_WORLD_SIZE=str(args.gpus),
                   MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), FMGAN_SELF_LAUNCHED='1')
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env,
                                      stdout=None if r == 0 else subprocess.DEVNULL))
    rc = 0
    try:
        alive = list(procs)
        while alive:
            for p in list(alive):
                code = p.poll()
                if code is None:
                    continue
                alive.remove(p)
                if code != 0 and rc == 0:
                    rc = code
                    for q in alive:          # a dead rank leaves the others in a collective: end exactly those PIDs
                        q.terminate()
            time.sleep(0.05)
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    return rc


def check_world(args, world):
    if world != args.gpus:
        raise SystemExit(f'bench.py: --gpus {args.gpus} but the job has {world} rank(s) (WORLD_SIZE='
                         f"{os.environ.get('WORLD_SIZE', 'unset')}): refusing to report a run of a different size")


def plumbing(args):
    """Everything of an N-rank bench run except the model: env rendezvous (gloo without GPUs), the shard of the global
    batch this rank owns, warm-up, barrier-bracketed timing, MAX over ranks, one JSON line on rank 0."""
    from Miscellaneous import distributed as D
    rank, world, device = D.init_distributed()
    check_world(args, world)
    wl = WORKLOADS[args.workload]
    batch = args.batch or wl['batch']
    lo, hi = D.shard_range(batch * world)
    assert hi - lo == batch
    x = torch.rand(batch, 64, 64)

    def step():
        time.sleep(0.002 * (rank + 1))          # ranks differ: the reported time must be the slowest rank's
        return x @ x

    dt = timed(step, args.steps, args.warmup, world)
    counts = D.all_gather((rank, lo, hi))
    if rank == 0:
        assert sorted(counts) == [(r, r * batch, (r + 1) * batch) for r in range(world)], counts
        print(json.dumps({'metric': '(photo,render) pairs/sec', 'value': world * batch * args.steps / dt, 'unit': 'pairs/s',
                          'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
                          'ms_per_step': 1e3 * dt / args.steps, 'higher_is_better': True, 'scaling': 'weak',
                          'vs_baseline': None, 'dtype': 'f32', 'data': 'plumbing test: no model, stand-in step',
                          'config': {'workload': f'{args.workload} (plumbing only)', 'pairs_per_gpu': batch,
                                     'global_pairs': batch * world, **dist_info(world)}}))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--workload', default='pairs1024', choices=sorted(WORKLOADS))
    ap.add_argument('--batch', type=int, default=0, help='pairs per GPU (default: the workload\'s)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-secondary', action='store_true')
    ap.add_argument('--graph', action='store_true', help='also time a HIP-graph replay of the forward and report it as '
                    'value (measured: no gain at these sizes, the step is GPU-bound)')
    ap.add_argument('--no-train', action='store_true', help='skip the training legs (train256, trainstep256) of the default run')
    ap.add_argument('--plumbing', action='store_true', help='no GPU work: a stand-in step through the same rendezvous, '
                    'sharding, barrier + MAX-over-ranks timing and JSON contract (CPU/gloo test of the N>1 launch path)')
    args = ap.parse_args()
    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        sys.exit(self_launch(args, sys.argv[1:]))
    if args.plumbing:
        return plumbing(args)

    warm_miopen_cache()
    import __graft_entry__
    # harness convenience on a fresh checkout (the product itself never builds): local rank 0 compiles, the others wait
    __graft_entry__.ensure_built(builder=int(os.environ.get('LOCAL_RANK', '0')) == 0)
    from Miscellaneous import distributed as D
    from op import _native
    # let MIOpen pick each encoder convolution's kernel by measurement during warm-up (pSp encoder 10.7 -> 9.9 ms).
    # Forward workloads only: the exhaustive search over every backward-data / weight-gradient solver of the training
    # workload (it times MIOpen's naive reference kernels too) takes more than 7 minutes of warm-up.
    torch.backends.cudnn.benchmark = not args.workload.startswith('train')
    # (MIOpen's measured find for the training convolutions was tried — FMGAN_TRAIN_FIND, round 3: one search pass over
    # the backward solvers of trainstep256 does not finish in 500 s and a second process searches again, so the
    # immediate-mode choice stays; the default run gives 48.9 pairs/s.)
    rank, world, device = D.init_distributed()
    assert torch.cuda.is_available(), 'bench.py needs a GPU'
    check_world(args, world)
    _native.lib()

    wl = WORKLOADS[args.workload]
    batch = args.batch or wl['batch']
    log(f'building models for {args.workload}')
    nets = build_models(wl['size'], device)
    if args.workload.startswith('train'):
        if args.workload == 'train256':
            step, inputs = make_train_step(nets, batch, device, rank, world)
        else:
            for m in nets.values():
                m.requires_grad_(True)
            step, trainer = make_trainstep(nets, batch, device, rank, world, wl['size'], wl.get('loss_nets', False),
                                           wl.get('precision', 'f32'))
        for i in range(args.warmup):
            step()
            torch.cuda.synchronize()
            log(f'{args.workload}: warm-up step {i + 1}/{args.warmup} done')
        it0 = trainer.iter_idx if args.workload.startswith('trainstep') else 0
        dt = timed(step, args.steps, 0, world)
        mix = None
        if args.workload.startswith('trainstep'):
            a = trainer.args
            its = range(it0, it0 + args.steps)
            mix = {'iterations': f'{it0}..{it0 + args.steps - 1}', 'd_steps': args.steps, 'g_steps': args.steps,
                   'r1_steps': sum(1 for i in its if i % a.d_reg_every == 0),
                   'path_length_steps': sum(1 for i in its if i % a.g_reg_every == 0),
                   'amortised': args.steps % TRAINSTEP_WINDOW == 0}
            torch.cuda.synchronize()
            mix['peak_hbm_gb'] = round(torch.cuda.max_memory_allocated() / 1e9, 1)
        if rank == 0:
            print(json.dumps({
                'metric': '(photo,render) pairs/sec (forward+backward)' if args.workload == 'train256' else
                          '(photo,render) pairs/sec (training iteration)', 'value': world * batch * args.steps / dt,
                'unit': 'pairs/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
                'ms_per_step': 1e3 * dt / args.steps, 'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
                'dtype': 'bf16 operands / f32 accumulate' if wl.get('precision') == 'bf16' else 'f32', 'data': 'synthetic',
                'config': {'workload': f"{args.workload}: {wl['desc']}", 'pairs_per_gpu': batch, 'global_pairs': batch * world,
                           'image_size': wl['size'], 'parallelism': f'dp{world} (DDP, 256 MiB buckets)' if world > 1 else 'single GPU',
                           'phase_mix': mix, **dist_info(world)}}))
        if world > 1:
            dist.barrier()
            dist.destroy_process_group()
        return
    step, inputs = make_step(nets, batch, device, rank)

    # headline kernel: the last (largest) blur of the workload; timed with HIP events inside the timed region
    r = wl['size']
    head = (batch * nets['g'].channels[r], r + 1, r + 1, r, r, 1, 1, 4)
    timer = LaunchTimer(lambda name, info: name == 'upfirdn2d' and info == head)
    for i in range(args.warmup):
        step()
        torch.cuda.synchronize()
        log(f'warm-up step {i + 1}/{args.warmup} done')
    _native.set_observer(timer)
    dt_eager = timed(step, args.steps, 0, world)      # eager launches; the headline kernel is bracketed by HIP events
    _native.set_observer(None)
    dt = dt_eager
    graphed = None
    if args.graph:
        from Util.hip_graph import GraphedForward
        graphed = GraphedForward(step.forward, inputs)
        dt = timed(lambda: graphed(*inputs), args.steps, 2, world)   # the same forward, replayed as one HIP graph
    pairs_per_s = world * batch * args.steps / dt
    log(f'{args.workload}: {pairs_per_s:.1f} pairs/s')
    # LABELLED variant, never `value`: the same forward with the modulated convs' fp32 contraction emulated on the bf16
    # matrix pipe (operands split into three bf16 pieces, six MFMAs per product, fp32 accumulation: fp32 accuracy, held to
    # the fp32 path's gates by tests/test_hip_modconv_bf16.py).  Timed with the same barriers right after the headline.
    x3 = None
    if args.workload == 'pairs1024' and not args.graph:
        def step_x3():
            with _native.modconv_precision('bf16x3'):
                return step()
        dt3 = timed(step_x3, args.steps, 3, world)
        x3 = {'pairs_per_s': world * batch * args.steps / dt3, 'ms_per_step': 1e3 * dt3 / args.steps,
              'what': 'pairs1024 with the 3x3 modulated convs on v_mfma_f32_32x32x16_bf16 via exact 3-way bf16 operand '
                      'splitting (6 MFMAs per product, fp32 accumulate; csrc/modconv_bf16.hip) - fp32-accurate, labelled, '
                      'not the headline'}
        log(f"pairs1024 with split-operand contraction: {x3['pairs_per_s']:.1f} pairs/s")

    out = {
        'metric': '(photo,render) pairs/sec', 'value': pairs_per_s, 'unit': 'pairs/s', 'n_gpus': world,
        'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': 1e3 * dt / args.steps, 'higher_is_better': True,
        'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
        'config': {'workload': f"{args.workload}: {wl['desc']}", 'pairs_per_gpu': batch, 'global_pairs': batch * world,
                   'image_size': wl['size'], 'parallelism': f'replicas x{world} (batch-sharded, no collective)',
                   'launch': 'eager' if graphed is None else 'hip-graph replay of the whole forward', **dist_info(world)},
        'eager': {'value': world * batch * args.steps / dt_eager, 'ms_per_step': 1e3 * dt_eager / args.steps},
    }
    if x3 is not None:
        out['pairs1024_bf16x3'] = x3

    # roofline of the dominant HBM kernel (upfirdn2d headline call), algorithmic bytes = 4*(in + out)
    summ = timer.summary()
    if (('upfirdn2d', head) in summ):
        ms, n = summ[('upfirdn2d', head)]
        # algorithmic bytes of the blur: input + output (the fused noise row adds 1/C of the output and is not counted)
        bytes_alg = 4.0 * head[0] * (head[1] * head[2] + head[3] * head[4])
        ach = bytes_alg / (ms * 1e-3) / 1e9
        # HBM traffic by PMC counters: the separate --pmc passes cannot run inside this process, so the committed
        # rocprofv3 result is quoted — but only while it still describes the kernel that just ran: the record names the
        # kernel source and its sha256 at measurement time (tools/headline_traffic.py writes it); after any edit of
        # that source the number is stale and `traffic` is null until the counters are collected again.
        traffic, traffic_src = None, None
        if batch == wl['batch'] and args.workload == 'pairs1024':
            traffic, traffic_src = committed_traffic()

        path = _native.BLUR_PATHS.get((head[0], head[1], head[2]))
        kname = {5: 'ufd_dmaring_f32<true> (blur + fused noise/bias/lrelu store, LDS-DMA row ring, path 1b)',
                 1: 'ufd_rowmarch_f32 (blur + fused noise/bias/lrelu store, register row-march, path 1)',
                 2: 'ufd_planetile_f32<true> (blur + fused epilogue, plane-tile)'}.get(path, f'upfirdn2d path {path}')
        out['roofline'] = {'bound': 'hbm', 'kernel': f'{kname} [{head[0]},{head[1]},{head[2]}]->[{head[0]},{head[3]},{head[4]}]',
                           'achieved': ach, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s', 'frac': ach / HBM_PEAK_GBS,
                           'traffic': traffic, 'traffic_source': traffic_src, 'avg_launch_ms': ms, 'launches': n,
                           'algorithmic_bytes': bytes_alg}

    # The reference-API op itself (op.upfirdn2d on a contiguous [B*C,2H+1,2W+1] tensor, no fused epilogue), standalone:
    # the in-situ kernel above additionally applies noise + bias + leaky ReLU in its store, which replaces a second
    # 2.1 GB elementwise pass but costs ~50 us of the blur's own time.
    if args.workload == 'pairs1024' and 'roofline' in out:
        from op import upfirdn2d as op_upfirdn2d
        g = nets['g']
        xin = torch.randn(batch, g.channels[r], r + 1, r + 1, device=device)
        kern = g.convs[-2].conv.blur.kernel
        for _ in range(3):
            op_upfirdn2d(xin, kern, pad=g.convs[-2].conv.blur.pad)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            op_upfirdn2d(xin, kern, pad=g.convs[-2].conv.blur.pad)
        e1.record()
        e1.synchronize()
        ms_op = e0.elapsed_time(e1) / 10
        ach_op = out['roofline']['algorithmic_bytes'] / (ms_op * 1e-3) / 1e9
        out['roofline_upfirdn2d_op'] = {'bound': 'hbm', 'kernel': 'op.upfirdn2d (contiguous input, no epilogue), 10 back-to-back launches',
                                        'achieved': ach_op, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s', 'frac': ach_op / HBM_PEAK_GBS,
                                        'avg_launch_ms': ms_op}
        del xin

    # per-kernel breakdown (separate, fully instrumented pass; not part of `value`)
    full = LaunchTimer(lambda name, info: True)
    _native.set_observer(full)
    step()
    torch.cuda.synchronize()
    _native.set_observer(None)
    fs = full.summary()
    tot = {}
    for (name, info), (ms, n) in fs.items():
        tot[name] = tot.get(name, 0.0) + ms * n
    conv = [(info, ms) for (name, info), (ms, n) in fs.items() if name == 'modconv2d']
    wino = [(info, ms) for (name, info), (ms, n) in fs.items() if name == 'modconv2d_winograd']
    if conv:
        def flops(i):
            b, cin, cout, h, w, mode = i
            return 2.0 * 9 * cin * cout * b * h * w
        best = max(conv, key=lambda t: flops(t[0]))
        tf = flops(best[0]) / (best[1] * 1e-3) / 1e12
        # all layers: ALGORITHMIC flops of the direct form over the time of whatever form ran them — the layers
        # served in Winograd form (16 instead of 36 products per 2x2 tile) can therefore read above the matrix peak
        tot_tf = sum(flops(i) for i, _ in conv + wino) / (sum(ms for _, ms in conv + wino) * 1e-3) / 1e12
        out['roofline_modconv'] = {'bound': 'mfma', 'kernel': f'modconv_mfma_f32 {best[0]} (largest layer on the direct MFMA kernel)',
                                   'achieved': tf, 'peak': FP32_MFMA_PEAK_TF, 'unit': 'TFLOP/s', 'frac': tf / FP32_MFMA_PEAK_TF,
                                   'all_layers_achieved': tot_tf,
                                   'direct_layers_achieved': sum(flops(i) for i, _ in conv) / (sum(ms for _, ms in conv) * 1e-3) / 1e12}
        if wino:
            out['roofline_modconv']['winograd_layers'] = {
                'count': len(wino), 'ms': round(sum(ms for _, ms in wino), 3),
                'algorithmic_tflops': sum(flops(i) for i, _ in wino) / (sum(ms for _, ms in wino) * 1e-3) / 1e12,
                'what': 'plain 3x3 layers served as F(2x2,3x3): own transform kernels + 16 batched fp32 GEMMs (hipBLASLt); '
                        'algorithmic flops of the direct form / time'}
    out['ms_by_op_instrumented'] = {k: round(v, 3) for k, v in sorted(tot.items())}

    if not args.no_secondary and args.workload == 'pairs1024':
        del nets, step, inputs
        torch.cuda.empty_cache()
        wl2 = WORKLOADS['pairs256']
        nets2 = build_models(wl2['size'], device)
        step2, in2 = make_step(nets2, wl2['batch'], device, rank)
        if not args.graph:
            dt2 = timed(step2, args.steps, args.warmup, world)
        else:
            g2 = GraphedForward(step2.forward, in2)
            dt2 = timed(lambda: g2(*in2), args.steps, 2, world)
            del g2
        log('pairs256 done')
        out['pairs_per_s_256'] = world * wl2['batch'] * args.steps / dt2
        out['ms_per_step_256'] = 1e3 * dt2 / args.steps
        out['config']['secondary'] = f"pairs256: {wl2['desc']}"
        # BASELINE config 2: synthesis network only, random W+ and input tensor, batch 32 @256^2
        g2net = nets2['g']
        gen2 = torch.Generator(device='cpu').manual_seed(4321 + rank)
        lat2 = torch.randn(wl2['batch'], g2net.n_latent, 512, generator=gen2).to(device)
        tsr2 = torch.randn(wl2['batch'], 512, 4, 4, generator=gen2).to(device)

        def synth_step():
            with torch.no_grad():
                return g2net(None, latent_styles=[lat2], input_is_latent=True, use_external_input_tensor=True,
                             external_input_tensor=tsr2)
        dt3 = timed(synth_step, args.steps, args.warmup, world)
        out['synthesis256_images_per_s'] = world * wl2['batch'] * args.steps / dt3
        out['config']['tertiary'] = 'cfg2: Generator(256) only, random W+ [32,14,512] and input tensor, fp32, B=32/GPU'
        del nets2, step2, g2net
        torch.cuda.empty_cache()
        if not args.no_train:
            # Training legs (BASELINE config 3 and a full train() iteration at 256^2, fp32), each as a child process
            # running this script with --workload: MIOpen has no pre-built kernels for gfx950 in this image and compiles
            # every backward convolution on first use — minutes on a fresh box (seconds once its cache is warm, see
            # warm_miopen_cache) — so the legs run inside a time box and the headline line is never held hostage by them.
            for name in ('train256', 'trainstep256', 'trainstep1024', 'trainstep1024_bf16'):
                # a training iteration is timed over whole windows of 16 (one R1 step + four path-length steps each)
                t_steps = TRAINSTEP_WINDOW if name.startswith('trainstep') else max(3, args.steps // 4)
                rec = train_leg(name, t_steps, world, float(os.environ.get('FMGAN_BENCH_TRAIN_TIMEOUT', '480')))
                if 'value' in rec:
                    out[name + '_pairs_per_s'] = rec['value']
                    out[name + '_ms_per_step'] = rec['ms_per_step']
                    out['config'][name] = f"{WORKLOADS[name]['desc']} ({t_steps} timed steps, child process)"
                    if rec['config'].get('phase_mix'):
                        out['config'][name + '_phase_mix'] = rec['config']['phase_mix']
                else:
                    out[name + '_pairs_per_s'] = None
                    out['config'][name] = rec['skipped']
        nets = build_models(wl['size'], device)
        _, inputs = make_step(nets, 1, device, rank)

    del graphed
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        log('cpu baseline (bounded sample of the CPU path)')
        out['cpu_baseline'] = cpu_baseline(nets, inputs, wl['size'])
        log('cpu baseline done')
    if rank == 0:
        print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
