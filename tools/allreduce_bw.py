#!/usr/bin/env python3
"""Gradient all-reduce micro-benchmark: bus bandwidth of the three gather_grad algorithms against the xGMI mesh.

    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        tools/allreduce_bw.py [--mb 64 256 1024] [--iters 20]

One rank per GPU (backend nccl = RCCL); with FMGAN_DIST_BACKEND=gloo it runs anywhere (plumbing check, numbers are
then host-memory numbers).  For a buffer of S bytes on N ranks an all-reduce moves 2*(N-1)/N*S per GPU:
    algbw = S / t            busbw = algbw * 2*(N-1)/N        (the nccl-tests convention)
xGMI on MI355X: 7 links per GPU x ~76 GB/s per direction (153 GB/s bidirectional).  A ring uses one link
direction per GPU (per-link bound: busbw <= ~76 GB/s per ring); the direct form (all-to-all of shards + local sum +
all-gather) drives all 7 links at once (bound ~ 7 x 76 = 532 GB/s).  The table reports busbw and its fraction of
7 x per-link.  Message sizes default to the training step's buckets: D ~115 MB, G + encoders ~1 GB in 256 MiB buckets.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, '3d-fm-gan_amd'))
os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402
from Miscellaneous import distributed as D  # noqa: E402

LINK_GBS_PER_DIRECTION = 76.5      # 153 GB/s bidirectional per xGMI link
LINKS = 7


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--mb', type=float, nargs='+', default=[16, 115, 256, 1024])
    ap.add_argument('--iters', type=int, default=20)
    args = ap.parse_args()
    rank, world, device = D.init_distributed()
    if world < 2:
        print('needs WORLD_SIZE >= 2 (launch with torch.distributed.run)')
        return
    gpu = device.type == 'cuda'
    rows = []
    for mb in args.mb:
        n = int(mb * (1 << 20) / 4) // world * world
        for algo in D.GRAD_ALGORITHMS:
            flat = torch.ones(n, dtype=torch.float32, device=device) * (rank + 1)
            for _ in range(3):
                D._reduce_flat(flat, world, algo)
            if gpu:
                torch.cuda.synchronize()
            dist.barrier()
            t0 = time.perf_counter()
            for _ in range(args.iters):
                D._reduce_flat(flat, world, algo)
            if gpu:
                torch.cuda.synchronize()
            dist.barrier()
            dt = (time.perf_counter() - t0) / args.iters
            t = torch.tensor([dt], dtype=torch.float64, device=device if dist.get_backend() == 'nccl' else 'cpu')
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = t.item()
            # correctness of the last result: repeated averaging of a constant converges to the mean (world+1)/2
            ok = abs(flat[0].item() - (world + 1) / 2) < 1e-3 and abs(flat[-1].item() - (world + 1) / 2) < 1e-3
            algbw = n * 4 / dt / 1e9
            busbw = algbw * 2 * (world - 1) / world
            rows.append(dict(bytes=n * 4, algorithm=algo, ms=dt * 1e3, algbw_GBs=algbw, busbw_GBs=busbw,
                             frac_of_7_links=busbw / (LINKS * LINK_GBS_PER_DIRECTION), ok=ok))
            del flat
    if rank == 0:
        print(f'# all-reduce bus bandwidth, {world} ranks, backend {dist.get_backend()}' + (f', {torch.cuda.get_device_name(0)}' if gpu else ''))
        print('| MiB | algorithm | ms | algbw GB/s | busbw GB/s | busbw / (7 x 76.5 GB/s) | result ok |')
        print('|---|---|---|---|---|---|---|')
        for r in rows:
            print(f"| {r['bytes'] / (1 << 20):.0f} | {r['algorithm']} | {r['ms']:.3f} | {r['algbw_GBs']:.1f} | {r['busbw_GBs']:.1f} | "
                  f"{r['frac_of_7_links']:.3f} | {r['ok']} |")
        print('JSON ' + json.dumps(rows))
    dist.barrier()
    dist.destroy_process_group()


if __name__ == '__main__':
    main()
