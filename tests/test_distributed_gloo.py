"""world_size-2 gloo tests of the data-parallel runtime that replaces nn.DataParallel / Miscellaneous/distributed.py.
Forward shards pairs with no collective; training averages gradients with bucketed all-reduce."""
import os
import sys

import pytest
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, q):
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
    q.put((rank, res))
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
    out = dict(q.get(timeout=500) for _ in range(2))
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
