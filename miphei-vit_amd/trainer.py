"""Minimal single-node trainer and the data-parallel gradient exchange (new capability: the reference is
single-device, ``Trainer(accelerator='gpu', devices=1)`` at /root/reference/src/train.py:205-207).

One process per GPU (``torch.distributed``; backend "nccl" is RCCL over xGMI on ROCm, "gloo" on CPU tests).
Tiles are independent, so the minibatch shards across ranks; the only exchange is the all-reduce of the flat
gradient buffer (6.7 M f32 for H-Optimus-0 + LoRA), split into two buckets: the decoder bucket is launched as soon
as the decoder backward has finished and overlaps the encoder backward on RCCL's own stream, the LoRA bucket follows.
"""
from __future__ import annotations

import torch
import torch.distributed as dist


class DataParallelSync:
    """Averages the flat gradient buffer over ranks; ``decoder_ready`` is called from inside ``engine.backward``."""

    def __init__(self, engine, group=None, force=False):
        self.engine, self.group = engine, group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.active = self.world > 1 or (force and dist.is_initialized())  # force: exercise the path on one rank
        self._work = None

    def broadcast_parameters(self, src=0):
        if self.active:
            fl = self.engine._ensure_flat()
            dist.broadcast(fl.flat, src=src, group=self.group)
            self.engine._pack_key = None

    def decoder_ready(self):
        if self.active:
            dec, _ = self.engine.grad_buckets()
            self._work = dist.all_reduce(dec, op=dist.ReduceOp.SUM, group=self.group, async_op=True)

    def finish(self):
        if self.active:
            dec, lora = self.engine.grad_buckets()
            w2 = dist.all_reduce(lora, op=dist.ReduceOp.SUM, group=self.group, async_op=True) if lora.numel() else None
            if self._work is not None:
                self._work.wait()
                self._work = None
            if w2 is not None:
                w2.wait()
            if self.world > 1:
                self.engine._flat.gflat.mul_(1.0 / self.world)


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
