"""Streams that really run side by side.

ROCm multiplexes HIP streams onto a small number of hardware queues (GPU_MAX_HW_QUEUES, 4 by
default), and PyTorch hands out streams from a pool; two streams that land on the same hardware
queue execute strictly one after the other.  Which streams share a queue depends on how many
streams the process created before, so the pipeline does not guess: it probes.  Two spin kernels
(`torch.cuda._sleep`) are queued on a pair of candidate streams; if the pair takes about as long as
one of them, the streams are on different hardware queues.
"""
import torch

_SPIN = 400000        # device cycles per probe kernel (~0.2 ms)


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
    pool = [torch.cuda.Stream(device=dev) for _ in range(candidates)]
    chosen, aliased = [], []
    for s in pool:
        clash = next((i for i, c in enumerate(chosen) if not overlaps(c, s, dev)), None)
        if clash is None and len(chosen) < n:
            chosen.append(s)
        elif clash is not None:
            aliased.append((s, clash))
    return chosen, aliased
