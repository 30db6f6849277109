"""One-process-per-GPU data parallelism over RCCL/xGMI (replaces the reference's nn.DataParallel wrapping,
train_3_encoder.py:355-362, and its dormant Miscellaneous/distributed.py helpers).

API kept: get_rank, synchronize, get_world_size, reduce_sum, gather_grad, all_gather, reduce_loss_dict
(Miscellaneous/distributed.py:18-135) — same arithmetic (SUM then / world; reduce of sorted, stacked losses to
rank 0).  Added: init_distributed() (env rendezvous; backend "nccl" is RCCL on ROCm, "gloo" on CPU),
Replica (exposes .module like DataParallel did) and shard_range().

Design for xGMI (7 point-to-point links per GPU): the forward path needs NO collective — each rank
runs the whole (photo, render) -> image stack on its own shard of pairs.  Training all-reduces gradients in a few
large flat buckets (default 256 MiB) instead of the reference's one all_reduce per parameter tensor
(distributed.py:66-75): E_W_Plus alone is ~1 GB of fp32 gradients in ~300 tensors, and per-link-bound rings want
few, large messages.  gather_grad() offers the library all-reduce, reduce-scatter + all-gather, and a DIRECT form
(all-to-all of shards + rank-ordered local sum + all-gather) that drives all 7 links of the full mesh at once;
tools/allreduce_bw.py measures their bus bandwidth against 7 x per-link.  No scaling curve has been measured yet: no
multi-GPU node was available to this build (rehearsals are gloo on CPU / one shared GPU).
"""
import os
import pickle

import torch
from torch import distributed as dist
from torch import nn


def _active():
    return dist.is_available() and dist.is_initialized()


def init_distributed(backend=None, force=False):
    """Join the job described by RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT (torchrun). Returns (rank, world, device).
    force: build the process group also for a job of one rank (a one-GPU box can then exercise the RCCL communicator,
    its collectives and DDP: tests/test_hip_train.py::test_rccl_group_of_one)."""
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    use_gpu = torch.cuda.is_available()
    if use_gpu:
        local = local % torch.cuda.device_count()     # rehearsals with more ranks than GPUs share devices (gloo only)
    backend = backend or os.environ.get('FMGAN_DIST_BACKEND')
    device = torch.device('cuda', local) if use_gpu else torch.device('cpu')
    if use_gpu:
        torch.cuda.set_device(device)
    if (world > 1 or force) and not _active():
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29500')
        backend = backend or ('nccl' if use_gpu else 'gloo')
        kw = {'device_id': device} if (use_gpu and backend == 'nccl') else {}
        dist.init_process_group(backend, rank=rank, world_size=world, **kw)
    return rank, world, device


def get_rank():
    return dist.get_rank() if _active() else 0


def get_world_size():
    return dist.get_world_size() if _active() else 1


def synchronize():
    if _active() and dist.get_world_size() > 1:
        dist.barrier()


def shard_range(total, rank=None, world=None):
    """Contiguous [lo, hi) slice of `total` units owned by `rank` (units = (photo, render) pairs)."""
    rank = get_rank() if rank is None else rank
    world = get_world_size() if world is None else world
    base, rem = divmod(total, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def reduce_sum(tensor):
    if not _active():
        return tensor
    tensor = tensor.clone()
    dist.all_reduce(tensor, op=dist.ReduceOp.SUM)
    return tensor


GRAD_ALGORITHMS = ('all_reduce', 'reduce_scatter', 'direct')


def _reduce_flat(flat, world, algorithm):
    """SUM over ranks then / world of one flat fp32 bucket whose length is a multiple of `world`, in place.

      all_reduce      one library all-reduce (RCCL picks ring / tree: on the 8-GPU xGMI mesh a ring moves 2*(7/8)*S per
                      GPU over ONE link direction).
      reduce_scatter  reduce_scatter_tensor + all_gather_into_tensor: the two halves of the all-reduce as separate
                      collectives, each rank owning 1/world of the bucket in between (the division happens on the
                      owned shard only: 1/world of the elementwise work).
      direct          the fully-connected mesh used as such: all_to_all of the world shards (every GPU sends S/world to
                      each peer over its own link, all 7 links busy at once), a local sum of the received shards in
                      RANK ORDER (bit-identical on every rank and from run to run, which ring orders are not), then
                      all_gather_into_tensor.  2*(S/world) per link and phase instead of the ring's 2*(7/8)*S.
    """
    if algorithm == 'all_reduce':
        dist.all_reduce(flat, op=dist.ReduceOp.SUM)
        flat.div_(world)
        return
    n = flat.numel() // world
    if algorithm == 'reduce_scatter':
        shard = torch.empty(n, dtype=flat.dtype, device=flat.device)
        dist.reduce_scatter_tensor(shard, flat, op=dist.ReduceOp.SUM)
    elif algorithm == 'direct':
        recv = torch.empty_like(flat)
        dist.all_to_all_single(recv, flat)
        shard = recv.view(world, n).sum(0)          # rank order: deterministic association
    else:
        raise ValueError(f'gather_grad: algorithm must be one of {GRAD_ALGORITHMS}')
    shard.div_(world)
    dist.all_gather_into_tensor(flat, shard)


def gather_grad(params, bucket_bytes=256 << 20, algorithm='reduce_scatter'):
    """Average gradients over ranks: SUM then / world — the arithmetic of the reference's gather_grad
    (distributed.py:66-75, one all_reduce per parameter tensor), bucketed into flat buffers of `bucket_bytes`.

    Every rank must bring the same buckets: a parameter that has a gradient on SOME rank but not on this one (its
    branch was not taken here) contributes zeros; a parameter without a gradient on every rank (the mapping network
    under input_is_latent=True) is skipped everywhere and stays `None`, as in the reference."""
    world = get_world_size()
    if world == 1:
        return
    params = [p for p in params if p.requires_grad]
    if not params:
        return
    dev = params[0].device
    has = torch.tensor([0 if p.grad is None else 1 for p in params], dtype=torch.int32, device=dev)
    dist.all_reduce(has, op=dist.ReduceOp.MAX)
    grads = []
    for p, h in zip(params, has.tolist()):
        if not h:
            continue
        if p.grad is None:
            p.grad = torch.zeros_like(p)
        grads.append(p.grad.data)
    bucket, size = [], 0

    def flush():
        nonlocal bucket, size
        if not bucket:
            return
        total = sum(g.numel() for g in bucket)
        padded = (total + world - 1) // world * world
        flat = torch.zeros(padded, dtype=bucket[0].dtype, device=bucket[0].device)
        off = 0
        for g in bucket:
            n = g.numel()
            flat[off:off + n].copy_(g.reshape(-1))
            off += n
        _reduce_flat(flat, world, algorithm)
        off = 0
        for g in bucket:
            n = g.numel()
            g.copy_(flat[off:off + n].view_as(g))
            off += n
        bucket, size = [], 0

    last_dtype = None
    for g in grads:
        if last_dtype is not None and g.dtype != last_dtype:
            flush()
        last_dtype = g.dtype
        bucket.append(g)
        size += g.numel() * g.element_size()
        if size >= bucket_bytes:
            flush()
    flush()


def all_gather(data):
    """Gather arbitrary picklable objects from every rank (distributed.py:78-110)."""
    world = get_world_size()
    if world == 1:
        return [data]
    out = [None] * world
    dist.all_gather_object(out, data)
    return out


def reduce_loss_dict(loss_dict):
    """Mean of each scalar loss on rank 0 (distributed.py:113-135): sorted keys -> stack -> reduce(dst=0) -> / world."""
    world = get_world_size()
    if world < 2:
        return loss_dict
    with torch.no_grad():
        keys = sorted(loss_dict.keys())
        losses = torch.stack([loss_dict[k] for k in keys], 0)
        dist.reduce(losses, dst=0)
        if dist.get_rank() == 0:
            losses /= world
        return {k: v for k, v in zip(keys, losses)}


class Replica(nn.Module):
    """Per-rank stand-in for nn.DataParallel(net): callers reach the network as `.module` (train_3_encoder.py:353,
    Util/network_util.py:317-318).  Forward is local; gradients are synchronised explicitly with gather_grad(),
    or wrap the module in torch's DistributedDataParallel, which also exposes `.module`."""

    def __init__(self, module):
        super().__init__()
        self.module = module

    def forward(self, *args, **kwargs):
        return self.module(*args, **kwargs)


def data_parallel(module, device=None, overlap=True, find_unused_parameters=True, single_rank_ddp=False):
    """DataParallel replacement: DDP (bucketed RCCL all-reduce overlapped with backward) when a process group is up
    and the module has trainable parameters, else a plain Replica.  find_unused_parameters defaults to True, the safe
    choice for an arbitrary module (DDP silently mis-reduces or stalls on parameters that receive no gradient): the
    3-encoder scheme leaves the Generator's mapping network and constant input unused (input_is_latent=True,
    use_external_input_tensor=True; Util/network_util.py:329-330).  Callers that know a network uses every parameter in
    every backward — the encoders with all W+ columns co-modulated, the discriminator — pass False and save the graph
    walk per iteration (train_3_encoder.Trainer, bench.py do)."""
    if device is not None:
        module = module.to(device)
    # modules that re-lay their conv weights for the GPU (pSp encoder: NHWC) do it now, so that DDP's bucket views are
    # built for the final parameter strides
    relayout = getattr(module, '_to_channels_last', None)
    if relayout is not None and getattr(module, 'channels_last', False) and next(module.parameters()).is_cuda:
        relayout()
    if (_active() and (get_world_size() > 1 or single_rank_ddp) and overlap and
            any(p.requires_grad for p in module.parameters())):
        ids = [device.index] if (device is not None and device.type == 'cuda') else None
        # broadcast_buffers=False: the only buffers are the encoders' BatchNorm statistics (eval mode, never updated,
        # SURVEY F13), the fixed noise maps and FIR taps — identical on every rank by construction
        return nn.parallel.DistributedDataParallel(module, device_ids=ids, bucket_cap_mb=256,
                                                   gradient_as_bucket_view=True, broadcast_buffers=False,
                                                   find_unused_parameters=find_unused_parameters)
    return Replica(module)
