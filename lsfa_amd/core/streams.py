"""Streams that really run side by side.

ROCm multiplexes HIP streams onto a small number of hardware queues (GPU_MAX_HW_QUEUES, 4 by
default), and PyTorch hands out streams from a pool; two streams that land on the same hardware
queue execute strictly one after the other.  Which streams share a queue depends on how many
streams the process created before, so the pipeline does not guess: it probes.  Two spin kernels
(`torch.cuda._sleep`) are queued on a pair of candidate streams; if the pair takes about as long as
one of them, the streams are on different hardware queues.
"""
import ctypes

import torch

_SPIN = 400000        # device cycles per probe kernel (~0.2 ms)
_OWNED = []           # hipStream_t handles created here and in use
_FREE = {}            # device index -> streams handed back by `release`, reused by the next `new_stream`
import os as _os
# Reuse is OFF by default: a hipStream that graphs were captured on carries PyTorch's BLAS workspace for (handle, stream); handing it to
# a later pipeline while library graphs of another pipeline replay made `torch.cuda.synchronize()` never return (two bf16 pipelines in one
# process, tests/test_graph_gpu.py::test_two_clips_interleaved_on_one_gpu_are_isolated; fine with LSFA_STREAM_REUSE=0, hung with 1).  A
# released stream is therefore only parked (a few hundred bytes of runtime state); what close() really gives back are the captured graphs
# and their private memory pools.
_REUSE = _os.environ.get('LSFA_STREAM_REUSE', '0') == '1'


def new_stream(device, high_priority=False):
    """A stream that is this caller's alone.  `torch.cuda.Stream()` draws from a pool of 32 per priority
    and wraps around: after enough pipelines / captures two Stream objects are the SAME hipStream_t.  PyTorch
    keys its BLAS workspace by (handle, stream) and captured GEMMs bake that address in, so graphs captured
    on such twins share split-K / stream-K scratch; replayed concurrently they corrupt each other, and the
    library kernels that spin on flags in that scratch never finish (observed: a device-side hang with two
    pipelines in one process).  Streams made here are created with hipStreamCreate and wrapped as
    ExternalStream; they live until `release`d (or the process ends)."""
    from lsfa_amd import hip
    dev = torch.device(device)
    idx = dev.index if dev.index is not None else torch.cuda.current_device()
    if _FREE.get(idx) and _REUSE and not high_priority:
        s = _FREE[idx].pop()
        _OWNED.append(s.cuda_stream)
        return s
    ptr = ctypes.c_void_p()
    with torch.cuda.device(dev):
        hip._check(hip.lib().lsfa_stream_create(ctypes.byref(ptr), ctypes.c_int(1 if high_priority else 0)), "lsfa_stream_create")
    s = torch.cuda.ExternalStream(ptr.value, device=dev)
    _OWNED.append(ptr.value)
    return s


def release(stream):
    """Hand a stream made by new_stream back (it must be idle, and the graphs captured on it dropped).  The handle is parked, not
    destroyed: PyTorch's caching allocator may still hold blocks that were `record_stream`ed on it and records an event on that
    stream when they are freed — on a destroyed stream that is an invalid-handle error.  Parked streams are reused by new_stream
    only with LSFA_STREAM_REUSE=1 (see _REUSE)."""
    ptr = stream.cuda_stream
    if ptr in _OWNED:
        _OWNED.remove(ptr)
        _FREE.setdefault(stream.device.index, []).append(stream)


def _pair_time(a, b, dev):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    main = torch.cuda.current_stream(dev)
    e0.record(main)
    for s in (a, b):
        s.wait_event(e0)
        with torch.cuda.stream(s):
            torch.cuda._sleep(_SPIN)
    for s in (a, b):
        main.wait_stream(s)
    e1.record(main)
    e1.synchronize()
    return e0.elapsed_time(e1)


def overlaps(a, b, dev):
    """True when kernels queued on streams a and b execute concurrently."""
    torch.cuda.synchronize(dev)
    _pair_time(a, a, dev)                        # warm
    serial = min(_pair_time(a, a, dev) for _ in range(2))       # two spins on ONE stream
    both = min(_pair_time(a, b, dev) for _ in range(2))
    return both < 0.75 * serial


def concurrent_streams(n, device, candidates=16):
    """Up to n mutually concurrent streams (fewer if the runtime has fewer hardware queues), plus the
    list of the other candidates tried, each tagged with the index of the chosen stream it aliases:
    -> (chosen, [(stream, alias_index), ...])."""
    dev = torch.device(device)
    pool = [new_stream(dev) for _ in range(candidates)]
    chosen, aliased, unused = [], [], []
    for s in pool:
        clash = next((i for i, c in enumerate(chosen) if not overlaps(c, s, dev)), None)
        if clash is None and len(chosen) < n:
            chosen.append(s)
        elif clash is not None and len(aliased) < n:
            aliased.append((s, clash))
        else:
            unused.append(s)
    torch.cuda.synchronize(dev)
    for s in unused:
        release(s)
    return chosen, aliased
