"""Minimal single-node trainer and the data-parallel gradient exchange (new capability: the reference is
single-device, ``Trainer(accelerator='gpu', devices=1)`` at /root/reference/src/train.py:205-207).

One process per GPU (``torch.distributed``; backend "nccl" is RCCL over xGMI on ROCm, "gloo" on CPU tests).
Tiles are independent, so the minibatch shards across ranks; the only exchange is the all-reduce of the flat
gradient buffer (6.7 M f32 for H-Optimus-0 + LoRA), split into buckets issued in reverse-forward order while the
backward pass runs: the decoder bucket as soon as the decoder backward has finished, then the LoRA gradients in
sub-buckets of ten blocks as the encoder backward walks from block 39 to block 0 (RCCL's own stream; only the last
sub-bucket can be exposed).
"""
from __future__ import annotations

import torch
import torch.distributed as dist


class DataParallelSync:
    """Averages the flat gradient buffer over ranks while the backward pass is still running.

    Buckets are contiguous slices of the flat gradient buffer, issued in the order the backward pass completes them
    (reverse-forward): the decoder bucket (``decoder_ready``, called from ``engine.backward`` when the
    decoder backward has finished, i.e. before the first encoder block) and ``lora_buckets`` sub-buckets of the LoRA
    gradients -- the flat layout is per block, so blocks ``hi..lo`` are one slice -- each issued by ``lora_block_done(l)``
    as soon as block ``lo`` of its range has produced its dA/dB.  Every all-reduce runs on the backend's own stream
    (RCCL's on ROCm), so only the last sub-bucket (``n_lora / lora_buckets`` floats; < 2 MB for H-Optimus-0 with 4) can be
    exposed.  ``finish`` waits for all of them on the compute stream (the 1/world average already rides on dL/d(out), see
    ``begin_step``; without it ``finish`` scales the buffer); with ``timing`` on it brackets that wait with events: the time the
    compute stream was held back by communication."""

    def __init__(self, engine, group=None, force=False, lora_buckets=4, timing=False, standin=None, hooks_only=False, decoder_split=1):
        self.engine, self.group = engine, group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.exchange = self.world > 1 or (force and dist.is_initialized())  # force: exercise the path on one rank
        # Pre-flight on a one-GPU box (bench.py --comm-standin): `standin = (blocks, usec)` launches csrc/standin.hip on a side
        # stream at every bucket-issue point -- workgroups that hold CUs the way a collective's ring kernels do -- and `finish`
        # waits for them; `hooks_only` keeps the five issue points (and the grouped LoRA launches they imply) without launching
        # anything: the A and B arms of the probe differ in the stand-in alone.
        self.standin = standin
        self.active = self.exchange or standin is not None or hooks_only
        self._side = None
        self.lora_buckets = max(1, int(lora_buckets))
        # decoder_split = K: the decoder bucket (18.9 MB for MIPHEI-ViT) leaves as K contiguous all-reduces instead of one.  A knob
        # for the first real multi-GPU run: a ring kernel holds its CUs for as long as its message lasts, and every wave-specialised
        # GEMM launch that overlaps it runs a second round; K smaller messages trade per-call latency for
        # shorter residency -- to be chosen from measurements together with NCCL_MAX_NCHANNELS / NCCL_PROTO (bench.py --gpus N).
        self.decoder_split = max(1, int(decoder_split))
        self.timing = timing
        self.exposed_events = []
        self.bucket_events = []       # timing: per step [(bytes, issue event, finish-start event, wait-done event), ...] in issue order
        self._work = []
        self._issued = []
        self._ranges = None
        self._prescaled = False       # this step's gradients were produced from dY / world (begin_step): finish() must not divide again

    def begin_step(self):
        """Factor for dL/d(out) of the step about to run: 1/world when the exchange is active, so that the SUM all-reduce of the
        buckets IS the average and ``finish`` has no pass over the gradient buffer left (round 5; ``ModelModule.training_step``
        calls this before ``engine.loss_and_grad``).  Callers that drive ``engine.backward`` themselves and never call it keep the
        divide-in-finish behaviour."""
        self._prescaled = bool(self.exchange and self.world > 1)
        return 1.0 / self.world if self._prescaled else 1.0

    def _buffers(self, kind):
        """flat parameter / gradient buffers of the engine (MIPHEI-ViT: one; UNETR baseline: decoder side + LoRA)"""
        get = getattr(self.engine, kind + "_buffers", None)
        if get is not None:
            return get()
        fl = self.engine._ensure_flat()
        return [fl.flat if kind == "param" else fl.gflat]

    def broadcast_parameters(self, src=0):
        if self.exchange:
            for buf in self._buffers("param"):
                dist.broadcast(buf, src=src, group=self.group)
            self.engine.params_changed()

    def _issue(self, t):
        if t.numel():
            if self.timing and t.is_cuda and self.exchange:
                ev = torch.cuda.Event(enable_timing=True)
                ev.record()                       # on the compute stream, at the point of the backward pass that releases the bucket
                self._issued.append((t.numel() * t.element_size(), ev))
            if self.standin is not None and t.is_cuda:
                from . import ops
                if self._side is None:
                    self._side = torch.cuda.Stream(device=t.device)
                self._side.wait_stream(torch.cuda.current_stream())     # ordered behind the bucket's producer, like the all-reduce
                with torch.cuda.stream(self._side):
                    ops.occupy_cus(int(self.standin[0]), int(self.standin[1]))
            if self.exchange:
                self._work.append(dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group, async_op=True))

    def decoder_ready(self):
        if self.active:
            dec, _ = self.engine.grad_buckets()
            k, n = self.decoder_split, dec.numel()
            if k <= 1 or n < k:
                self._issue(dec)
            else:
                edges = [round(i * n / k) for i in range(k + 1)]
                for a, b in zip(edges[:-1], edges[1:]):
                    self._issue(dec[a:b])

    def _lora_ranges(self):
        """{lowest block of a sub-bucket: slice of the LoRA gradient region}; blocks are contiguous in the flat layout."""
        if self._ranges is None:
            _, lora = self.engine.grad_buckets()
            L = self.engine.lora_blocks()
            per = lora.numel() // L if L else 0
            nb = min(self.lora_buckets, L) if L else 0
            edges = [round(i * L / nb) for i in range(nb + 1)] if nb else []
            self._ranges = {edges[i]: (edges[i] * per, edges[i + 1] * per) for i in range(nb)}
        return self._ranges

    def lora_block_done(self, l):
        """Called by ``engine._encoder_bwd`` after block ``l`` (descending) has written its LoRA gradients."""
        if self.active:
            r = self._lora_ranges().get(l)
            if r is not None:
                _, lora = self.engine.grad_buckets()
                self._issue(lora[r[0]:r[1]])

    def finish(self):
        if not self.active:
            return
        if self.timing:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
        marks = []
        for wk in self._work:
            wk.wait()
            if self.timing and self._issued:
                ev = torch.cuda.Event(enable_timing=True)
                ev.record()
                marks.append(ev)
        if self.timing and self._issued and len(marks) == len(self._issued):
            self.bucket_events.append([(nb, iss, e0, mk) for (nb, iss), mk in zip(self._issued, marks)])
        self._issued = []
        self._work = []
        if self._side is not None:
            torch.cuda.current_stream().wait_stream(self._side)        # the stand-in kernels, like the all-reduces, end before clip + Adam
        if self.world > 1 and not self._prescaled:
            for buf in self._buffers("grad"):
                buf.mul_(1.0 / self.world)
        self._prescaled = False
        if self.timing:
            e1.record()
            self.exposed_events.append((e0, e1))

    def exposed_ms(self):
        """Mean per-step time the compute stream waited in ``finish`` (all-reduce tail + the 1/world scale)."""
        if not self.exposed_events:
            return None
        torch.cuda.synchronize()
        ms = [a.elapsed_time(b) for a, b in self.exposed_events]
        self.exposed_events = []
        return sum(ms) / len(ms)


    def bucket_report(self):
        """Per bucket (issue order: decoder, then the LoRA sub-buckets), averaged over the timed steps: bytes, ``slack_ms`` = time
        between the bucket's issue inside the backward pass and the start of ``finish`` (what the exchange could hide behind) and
        ``wait_ms`` = how long the compute stream then stalled for it (cumulative waits: bucket k's stall starts where bucket
        k-1's ended).  Explains a scaling number: efficiency is lost exactly where wait_ms is not ~0."""
        if not self.bucket_events:
            return None
        torch.cuda.synchronize()
        n = len(self.bucket_events[0])
        steps = [st for st in self.bucket_events if len(st) == n]
        out = []
        for k in range(n):
            slack = [st[k][1].elapsed_time(st[k][2]) for st in steps]
            wait = [(st[k - 1][3] if k else st[k][2]).elapsed_time(st[k][3]) for st in steps]
            out.append({"bytes": int(steps[0][k][0]), "slack_ms": round(sum(slack) / len(slack), 4),
                        "wait_ms": round(sum(wait) / len(wait), 4)})
        self.bucket_events = []
        return out


def allreduce_mean_(flat, world, group=None):
    """Reference semantics of the exchange on any backend (used by the gloo CPU tests)."""
    if world > 1:
        dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
        flat.mul_(1.0 / world)
    return flat


def fit(module, batches, total_iters=None):
    """Drive ``ModelModule.training_step`` over an iterable of {"image","target"} batches."""
    if total_iters is not None:
        module.total_iters = total_iters
    losses = []
    for i, batch in enumerate(batches):
        losses.append(module.training_step(batch, i))
    return losses


def predict(module, batches):
    module.generator.eval()
    outs = []
    with torch.no_grad():
        for i, batch in enumerate(batches):
            outs.append(module.predict_step(batch, i))
    return outs
