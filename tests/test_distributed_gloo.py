"""world_size-2 gloo tests of the data-parallel runtime that replaces nn.DataParallel / Miscellaneous/distributed.py.
Forward shards pairs with no collective; training averages gradients with bucketed all-reduce."""
import os
import sys

import pytest
import numpy as np
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _toy_modules():
    import synth
    lin, lin2, dead = torch.nn.Linear(5, 3), torch.nn.Linear(3, 2), torch.nn.Linear(2, 2)
    for name, m in (('lin', lin), ('lin2', lin2), ('dead', dead)):
        m.load_state_dict({k: synth.tensor(f'toy/{name}/{k}', v.shape) for k, v in m.state_dict().items()})
    return lin, lin2, dead


def _by_value(v):
    if torch.is_tensor(v):
        return v.detach().cpu().numpy()
    if isinstance(v, dict):
        return {k: _by_value(x) for k, x in v.items()}
    if isinstance(v, (tuple, list)):
        return type(v)(_by_value(x) for x in v)
    return v


def _tensors(v):
    if isinstance(v, np.ndarray):
        return torch.from_numpy(v)
    if isinstance(v, dict):
        return {k: _tensors(x) for k, x in v.items()}
    if isinstance(v, (tuple, list)):
        return type(v)(_tensors(x) for x in v)
    return v


def _worker(rank, world, port, q):
    try:
        _worker_body(rank, world, port, q)
    except BaseException:
        import traceback
        q.put((rank, {'error': traceback.format_exc()}))      # fail the parent at once instead of at its queue timeout
        raise


def _worker_body(rank, world, port, q):
    sys.path.insert(0, os.path.join(ROOT, '3d-fm-gan_amd'))
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR='127.0.0.1',
                      MASTER_PORT=str(port))
    torch.set_num_threads(2)
    from Miscellaneous import distributed as D
    import resnet_encoder
    import synth
    r, w, device = D.init_distributed(backend='gloo')
    assert (r, w) == (rank, world) and D.get_rank() == rank and D.get_world_size() == world
    D.synchronize()
    res = {}
    # shard_range partitions the pairs with no overlap
    res['shard'] = D.shard_range(7)
    # reduce_sum / all_gather / reduce_loss_dict semantics (distributed.py:53-63, 78-110, 113-135)
    res['reduce_sum'] = D.reduce_sum(torch.tensor([1.0 + rank])).item()
    res['all_gather'] = D.all_gather({'rank': rank, 'payload': 'x' * (rank + 1)})
    red = D.reduce_loss_dict({'b': torch.tensor(2.0 * (rank + 1)), 'a': torch.tensor(1.0 * (rank + 1))})
    res['loss'] = {k: v.item() for k, v in red.items()}
    # gradient averaging == single-process gradient on the concatenated batch (BN in eval mode)
    net = resnet_encoder.resnet18(tensor_encoding=False)
    net.load_state_dict(synth.state_dict('resnet', net.state_dict(), seed=5))
    net.eval()
    x = synth.tensor('ddp/x', (4, 3, 64, 64), dist='uniform')
    lo, hi = D.shard_range(4)
    loss = net(x[lo:hi]).pow(2).mean()
    loss.backward()
    D.gather_grad(net.parameters(), bucket_bytes=1 << 20)    # several buckets
    res['grad'] = torch.cat([p.grad.reshape(-1) for p in net.parameters()])[::997].clone()
    # DDP wrapper exposes .module and produces the same averaged gradients
    net2 = resnet_encoder.resnet18(tensor_encoding=False)
    net2.load_state_dict(synth.state_dict('resnet', net2.state_dict(), seed=5))
    net2.eval()
    ddp = D.data_parallel(net2)
    assert hasattr(ddp, 'module') and ddp.module is net2
    ddp(x[lo:hi]).pow(2).mean().backward()
    res['grad_ddp'] = torch.cat([p.grad.reshape(-1) for p in net2.parameters()])[::997].clone()
    frozen = D.data_parallel(torch.nn.Linear(2, 2).requires_grad_(False))
    assert isinstance(frozen, D.Replica)
    # every gather_grad algorithm (library all-reduce / reduce-scatter + all-gather / direct all-to-all form) with a
    # parameter set that differs between ranks: lin2 has a gradient on rank 0 only (zeros are substituted on rank 1 so
    # both ranks bring the same buckets), `dead` has none anywhere (stays None, as in the reference's gather_grad)
    for algo in D.GRAD_ALGORITHMS:
        lin, lin2, dead = _toy_modules()
        xx = synth.tensor('gg/x', (4, 5))
        y = lin(xx[rank * 2:(rank + 1) * 2])
        loss = y.pow(2).mean()
        if rank == 0:
            loss = loss + lin2(y).sum()
        loss.backward()
        params = list(lin.parameters()) + list(lin2.parameters()) + list(dead.parameters())
        D.gather_grad(params, bucket_bytes=40, algorithm=algo)          # 40 bytes: several, ragged buckets
        assert dead.weight.grad is None and dead.bias.grad is None
        res['gg_' + algo] = torch.cat([p.grad.reshape(-1) for p in list(lin.parameters()) + list(lin2.parameters())]).clone()
    # differentiable global mean used by the path-length regulariser (train_3_encoder.py::_global_mean)
    import train_3_encoder as T
    v = (synth.tensor('gm/v', (4,))[rank * 2:(rank + 1) * 2]).clone().requires_grad_(True)
    m = T._global_mean(v)
    (m * m).backward()
    res['gmean'] = (m.item(), v.grad.clone())
    # pSp encoder with NHWC-laid-out conv weights under DDP (round-1 failure: 'Grad strides do not match bucket view
    # strides' for the head convs when the relayout happened after DDP construction)
    import types
    import warnings
    from psp_encoder_model.encoders import psp_encoders
    enc = psp_encoders.GradualStyleEncoder(18, 'ir_se', types.SimpleNamespace(input_nc=3, n_styles=1))
    enc.load_state_dict(synth.state_dict('psp', enc.state_dict(), seed=7))
    enc.eval()
    enc._to_channels_last()
    assert enc.styles[0].convs[0].weight.is_contiguous(memory_format=torch.channels_last)
    with warnings.catch_warnings(record=True) as caught:
        warnings.simplefilter('always')
        ddp_enc = D.data_parallel(enc)
        xe = synth.tensor('ddp/xe', (2, 3, 64, 64), dist='uniform')
        ddp_enc(xe[rank:rank + 1]).pow(2).mean().backward()
    # [O,I,1,1] weights are contiguous in BOTH memory formats (their strides are ambiguous, the bytes identical): for
    # those the stride comparison is noise.  Any kernel larger than 1x1 must match the bucket view exactly.
    import re
    bad = [str(w.message) for w in caught if 'Grad strides do not match bucket view strides' in str(w.message)
           and not re.search(r'grad\.sizes\(\) = \[\d+, \d+, 1, 1\]', str(w.message))]
    assert not bad, bad[0]
    # (one style head: the pyramid's lateral layers are unused and have no gradient)
    res['grad_psp'] = torch.cat([p.grad.reshape(-1) for p in enc.parameters() if p.grad is not None])[::4999].clone()
    # tensors cross the queue BY VALUE (numpy): a torch tensor travels as a shared-memory file that the parent opens
    # lazily, and a worker that has already exited by then leaves it a FileNotFoundError
    q.put((rank, _by_value(res)))
    D.synchronize()
    torch.distributed.destroy_process_group()


@pytest.mark.timeout(600)
def test_world_size_2_gloo():
    sys.path.insert(0, os.path.join(ROOT, '3d-fm-gan_amd'))
    import resnet_encoder
    import synth
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    out = {}
    for _ in range(2):
        r, res = q.get(timeout=500)
        assert 'error' not in res, res.get('error')
        out[r] = _tensors(res)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert out[0]['shard'] == (0, 4) and out[1]['shard'] == (4, 7)
    assert out[0]['reduce_sum'] == out[1]['reduce_sum'] == 3.0
    assert [d['rank'] for d in out[0]['all_gather']] == [0, 1] and out[1]['all_gather'][1]['payload'] == 'xx'
    assert out[0]['loss'] == {'a': 1.5, 'b': 3.0}            # mean over ranks, on rank 0 only
    # reference gradient: whole batch in one process; mean over 4 samples == mean of the two per-rank means
    net = resnet_encoder.resnet18(tensor_encoding=False)
    net.load_state_dict(synth.state_dict('resnet', net.state_dict(), seed=5))
    net.eval()
    x = synth.tensor('ddp/x', (4, 3, 64, 64), dist='uniform')
    net(x).pow(2).mean().backward()
    ref = torch.cat([p.grad.reshape(-1) for p in net.parameters()])[::997]
    for r in (0, 1):
        torch.testing.assert_close(out[r]['grad'], ref, atol=1e-5, rtol=1e-4)
        torch.testing.assert_close(out[r]['grad_ddp'], ref, atol=1e-5, rtol=1e-4)
    torch.testing.assert_close(out[0]['grad'], out[1]['grad'], atol=0, rtol=0)
    # gather_grad algorithms vs the single-process gradient of mean-over-ranks of the per-rank losses
    lin, lin2, dead = _toy_modules()
    xx = synth.tensor('gg/x', (4, 5))
    y0, y1 = lin(xx[:2]), lin(xx[2:])
    (0.5 * (y0.pow(2).mean() + lin2(y0).sum()) + 0.5 * y1.pow(2).mean()).backward()
    ref = torch.cat([p.grad.reshape(-1) for p in list(lin.parameters()) + list(lin2.parameters())])
    from Miscellaneous import distributed as D
    for algo in D.GRAD_ALGORITHMS:
        for r in (0, 1):
            torch.testing.assert_close(out[r]['gg_' + algo], ref, atol=1e-6, rtol=1e-5)
        assert torch.equal(out[0]['gg_' + algo], out[1]['gg_' + algo])           # every rank ends with the same bits
    # global mean and its gradient: m = mean of all 4 values, d(m^2)/dv_i = 2m/4 summed over the 2 ranks' losses
    v = synth.tensor('gm/v', (4,))
    for r in (0, 1):
        m, g = out[r]['gmean']
        assert abs(m - v.mean().item()) < 1e-6
        torch.testing.assert_close(g, torch.full((2,), 2 * (2 * v.mean().item()) / 4), atol=1e-6, rtol=1e-5)
    # pSp encoder, channels_last weights, DDP == single process on both samples
    import types
    from psp_encoder_model.encoders import psp_encoders
    enc = psp_encoders.GradualStyleEncoder(18, 'ir_se', types.SimpleNamespace(input_nc=3, n_styles=1))
    enc.load_state_dict(synth.state_dict('psp', enc.state_dict(), seed=7))
    enc.eval()
    xe = synth.tensor('ddp/xe', (2, 3, 64, 64), dist='uniform')
    (0.5 * enc(xe[:1]).pow(2).mean() + 0.5 * enc(xe[1:]).pow(2).mean()).backward()
    ref = torch.cat([p.grad.reshape(-1) for p in enc.parameters() if p.grad is not None])[::4999]
    assert ref.numel() > 1000
    for r in (0, 1):
        torch.testing.assert_close(out[r]['grad_psp'], ref, atol=1e-5 * float(ref.abs().max()), rtol=1e-3)


@pytest.mark.timeout(300)
def test_bench_two_rank_launch_plumbing():
    """`python -m torch.distributed.run ... bench.py --gpus 2` exactly as the driver launches it, with the model replaced
    by a stand-in step (--plumbing; gloo, no GPU): rendezvous on 127.0.0.1, shard_range of the global batch, barriers,
    MAX over ranks of the timed region (rank 1 sleeps twice as long as rank 0), ONE JSON line from rank 0."""
    import json
    import subprocess
    port = 29500 + ((os.getpid() + 977) % 2000)
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr',
           '127.0.0.1', '--master-port', str(port), os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '5',
           '--warmup', '1', '--plumbing']
    proc = subprocess.run(cmd, capture_output=True, text=True, timeout=280, cwd=ROOT)
    assert proc.returncode == 0, proc.stderr[-2000:]
    lines = [ln for ln in proc.stdout.splitlines() if ln.startswith('{')]
    assert len(lines) == 1, proc.stdout
    rec = json.loads(lines[0])
    for k in ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling',
              'vs_baseline', 'dtype', 'data', 'config'):
        assert k in rec, k
    assert rec['n_gpus'] == 2 and rec['steps'] == 5 and rec['scaling'] == 'weak'
    assert rec['config']['global_pairs'] == 2 * rec['config']['pairs_per_gpu']
    assert rec['ms_per_step'] >= 4.0            # the slowest rank's 4 ms sleep, not rank 0's 2 ms
    assert abs(rec['value'] - rec['config']['global_pairs'] / (rec['ms_per_step'] * 1e-3)) < 1e-6 * rec['value']


@pytest.mark.timeout(300)
def test_bench_launches_its_own_ranks():
    """Plain `python bench.py --gpus 2 --plumbing`, no external launcher and no WORLD_SIZE: the parent starts the two
    ranks itself and relays rank 0's line, which must say n_gpus == 2 with the rank count the process group reports."""
    import json
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_ADDR', 'MASTER_PORT')}
    proc = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '5', '--warmup', '1',
                           '--plumbing'], capture_output=True, text=True, timeout=280, cwd=ROOT, env=env)
    assert proc.returncode == 0, proc.stderr[-2000:]
    lines = [ln for ln in proc.stdout.splitlines() if ln.startswith('{')]
    assert len(lines) == 1, proc.stdout
    rec = json.loads(lines[0])
    assert rec['n_gpus'] == 2 and rec['config']['ranks'] == 2 and rec['config']['backend'] == 'gloo'
    assert rec['config']['launcher'] == 'bench.py self-launch'
    assert rec['ms_per_step'] >= 4.0


@pytest.mark.timeout(120)
def test_bench_refuses_a_rank_count_other_than_gpus():
    """--gpus 2 inside a job of ONE rank (WORLD_SIZE=1) must fail instead of printing a 1-GPU line labelled otherwise."""
    import subprocess
    env = dict(os.environ, WORLD_SIZE='1', RANK='0', LOCAL_RANK='0')
    proc = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '2', '--warmup', '0',
                           '--plumbing'], capture_output=True, text=True, timeout=100, cwd=ROOT, env=env)
    assert proc.returncode != 0
    assert 'refusing' in proc.stderr and not [ln for ln in proc.stdout.splitlines() if ln.startswith('{')]


def test_single_process_helpers_are_noops():
    sys.path.insert(0, os.path.join(ROOT, '3d-fm-gan_amd'))
    from Miscellaneous import distributed as D
    assert D.get_rank() == 0 and D.get_world_size() == 1
    D.synchronize()
    t = torch.ones(3)
    assert D.reduce_sum(t) is t
    assert D.all_gather('x') == ['x']
    d = {'a': torch.tensor(1.0)}
    assert D.reduce_loss_dict(d) is d
    assert D.shard_range(10, 1, 4) == (3, 6)
    D.gather_grad([torch.nn.Parameter(torch.ones(2))])
