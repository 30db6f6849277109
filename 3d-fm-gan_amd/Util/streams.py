"""Side HIP streams for independent branches of the forward (inference only).

The path mixes MFMA-bound kernels (3x3 modulated convs, encoder convs) with HBM-bound ones (ToRGB, skip upsample,
blur, activation) and small launches that cannot fill 256 CUs alone; branches with no data dependency are issued on
side streams so the hardware can co-schedule them (two processes sharing one MI355X measured +13 % pairs/s, which is
the headroom this recovers inside one process).  Fork/join are stream waits only — no host synchronisation — so the
forward stays capturable in a HIP graph.
"""
import os

import torch

_POOL = {}
ENABLED = os.environ.get('FMGAN_NO_OVERLAP', '0') != '1'   # set False (or FMGAN_NO_OVERLAP=1) for single-stream issue


def overlap_ok(t):
    return ENABLED and t.is_cuda and not torch.is_grad_enabled()


def side_streams(device, n, group='default'):
    """n side streams of a named group (groups do not share streams: work queued for one branch never sits in front
    of another branch's launches)."""
    key = (device.type, device.index, group)
    pool = _POOL.setdefault(key, [])
    while len(pool) < n:
        pool.append(torch.cuda.Stream(device=device))
    return pool[:n]


def _record(obj, stream):
    if torch.is_tensor(obj):
        obj.record_stream(stream)
    elif isinstance(obj, (tuple, list)):
        for o in obj:
            _record(o, stream)


def run_on(stream, fn, *args):
    """Run fn(*args) on `stream` after everything already queued on the current stream.  Returns (join, result):
    call join() before the current stream consumes `result`."""
    main = torch.cuda.current_stream(stream.device)
    stream.wait_stream(main)
    for a in args:
        _record(a, stream)            # inputs were allocated on `main`: keep them alive for `stream`
    with torch.cuda.stream(stream):
        out = fn(*args)

    def join():
        main.wait_stream(stream)
        _record(out, main)            # outputs were allocated on `stream`: they are consumed on `main`

    return join, out


def run_deferred(stream, fn, *args):
    """Like run_on, but the join is an event recorded right after fn's launches: wait() makes the CURRENT stream wait
    for this call only, not for whatever is queued on `stream` afterwards.  Returns (wait, result)."""
    main = torch.cuda.current_stream(stream.device)
    stream.wait_stream(main)
    for a in args:
        _record(a, stream)
    with torch.cuda.stream(stream):
        out = fn(*args)
        done = torch.cuda.Event()
        done.record(stream)

    def wait():
        cur = torch.cuda.current_stream(stream.device)
        cur.wait_event(done)
        _record(out, cur)

    return wait, out
