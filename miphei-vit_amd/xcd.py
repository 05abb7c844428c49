"""Per-XCD speed ranking (round 5).  Under a dense bf16 MFMA load the eight XCDs of an MI355X settle at clocks several per cent apart
-- stable over a run, different from box to box (tools/ws_timing.py, docs/rounds/round5.md) -- and a statically partitioned launch ends
with its slowest XCD.  Where a launch has slack to place (the band items of a ragged wave-specialised GEMM: 192 items for 256 CUs at
batch 16), the library can place it by this ranking.  OPT-IN (MIPHEI_XCD_RANK=1, then measured once per device and process, ~2 ms):
on the boxes of round 5 the XCDs were only +-2 % apart under this probe and the ranking was within its own noise; what pays is the
XCD-contiguous placement the library uses by default (csrc/gemm_ws.hip: fc1 + SwiGLU 122.5 -> 117.5 us)."""
from __future__ import annotations

import ctypes as C
import os

import torch

from . import _lib as L

_done: dict = {}


def measure(device=None, iters=2000, repeats=5):
    """mean duration (10 ns ticks) of the probe workgroups per XCD, fastest first ranking: (durations[8], rank[8])"""
    dev = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
    cus = torch.cuda.get_device_properties(dev).multi_processor_count
    with torch.cuda.device(dev):
        out = torch.zeros(cus, device=dev, dtype=torch.int64)
        acc = torch.zeros(8, dtype=torch.float64)
        stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        for r in range(repeats + 1):
            L.check(L.lib().mvit_xcd_probe(C.c_void_p(out.data_ptr()), cus, iters, stream), "mvit_xcd_probe")
            torch.cuda.synchronize()
            if r == 0:
                continue            # (clock ramp of the first launch)
            t = out.cpu().double()
            for x in range(8):
                acc[x] += t[x::8].mean()
    dur = (acc / repeats).tolist()
    order = sorted(range(8), key=lambda x: dur[x])       # fastest XCD first
    rank = [0] * 8
    for r, x in enumerate(order):
        rank[x] = r
    return dur, rank


def apply(device=None, force=False):
    """measure and install the ranking for the device (once per process); returns the ranking or None when switched off"""
    if os.environ.get("MIPHEI_XCD_RANK", "0") != "1" and not force:
        return None
    dev = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
    key = dev.index
    if key in _done and not force:
        return _done[key]
    dur, rank = measure(dev)
    with torch.cuda.device(dev):
        arr = (C.c_int * 8)(*rank)
        L.check(L.lib().mvit_set_xcd_rank(arr), "mvit_set_xcd_rank")
    _done[key] = rank
    return rank
