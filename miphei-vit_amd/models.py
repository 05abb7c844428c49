"""``ModelModule``: the training / prediction step around the generator.

Mirrors the constructor and step methods of the reference LightningModule (``/root/reference/src/models.py``:
``__init__`` :22-73, ``forward`` :75-76, ``predict_step`` :78-79, ``training_step`` :87-143 non-GAN branch,
``configure_optimizers`` :348-371).  pytorch_lightning is optional: when it is importable the class derives from
``LightningModule`` (manual optimisation, as the reference); otherwise from ``nn.Module`` and ``fit`` / ``predict`` in
``trainer.py`` drive it.

The non-GAN ``training_step`` runs as ONE fused device sequence: HIP generator forward -> fused WeightedMSE loss +
gradient -> HIP backward -> (optional) RCCL all-reduce of the two gradient buckets, the decoder bucket overlapped
with the encoder backward -> global-norm clip + Adam on the flat parameter buffer -> LR schedule.  The reference's
per-step host sync (``torch.isnan(fake).any()``, models.py:102-105) becomes a device-side gate plus an asynchronous host
check: the Adam kernel skips the update (and every later one) when the gradient norm is non-finite and raises a sticky
device flag; the host copies that flag out without blocking after every step, inspects the copies that have landed and
then saves ``weights_nan.ckpt`` -- still the last finite weights -- and raises ``ValueError("Nan found")``.
"""
from __future__ import annotations

import math

import torch
import torch.nn as nn

from .loss import WeightedMSELoss
from .metrics import PixMetrics
from .utils import pix2pix_lr_scheduler

try:  # pragma: no cover - not installed in the build container
    import pytorch_lightning as pl
    _Base = pl.LightningModule
except Exception:  # noqa: BLE001
    pl = None
    _Base = nn.Module


class ModelModule(_Base):
    def __init__(self, generator, discriminator, lr_g, lr_d, loss_reconstruct, cell_metrics=None, cell_loss=None,
                 gan_train=False):
        super().__init__()
        if gan_train or discriminator is not None and gan_train:
            raise NotImplementedError("the PatchGAN branch is outside the MI355X hot path (gan_train: false is the "
                                      "MIPHEI-ViT default)")
        if cell_loss is not None:
            raise NotImplementedError("the cell-level LOSS (reference models.py:236-240, a training objective through "
                                      "MeanCellExtrator) is outside the MI355X hot path; cell_metrics (validation) is supported")
        if cell_metrics is not None and not callable(getattr(cell_metrics, "update", None)):
            raise TypeError("cell_metrics must expose update(preds, nuclei_masks, slide_names) (cells.CellMetrics)")
        self.generator = generator
        self.foreground_head = hasattr(generator, "foreground_head")
        self.discriminator = None
        self.gan_train = False
        self.automatic_optimization = False
        # validation-time per-nucleus statistics (reference models.py:54-61, 233-241): the segmented-reduction extractor of
        # csrc/cells.hip behind cells.CellMetrics.update; its sklearn-side ``compute`` stays evaluation code
        self.use_cell_metrics = cell_metrics is not None
        self.cell_metrics = cell_metrics
        self.cell_loss = None
        if self.use_cell_metrics and hasattr(cell_metrics, "marker_pred_cols") and hasattr(cell_metrics, "marker_cols"):
            self.logreg_layer = nn.Linear(len(cell_metrics.marker_pred_cols), len(cell_metrics.marker_cols))
        self.lr_g, self.lr_d = lr_g, lr_d
        self.loss_reconstruct = loss_reconstruct
        self.vit_lr_decay = False
        # fused-step state
        self.total_iters = None
        self.global_step_ = 0
        self.nan_check = True         # asynchronous host side of the NaN guard (the device side is always on)
        self._pending = []            # (event, pinned host int32) copies of the device NaN flag, oldest first
        self.grad_sync = None  # set by trainer.DataParallelSync for multi-GPU runs
        self.last_loss = None
        # PSNR / SSIM collections of the reference (models.py:35-52), device-resident state, updated after every step
        self.train_pix_metrics = PixMetrics((-0.9, 0.9))
        self.val_pix_metrics = self.train_pix_metrics.clone(prefix="val_")
        self.test_pix_metrics = self.train_pix_metrics.clone(prefix="test_")
        self.update_pix_metrics = True

    # ------------------------------------------------------------------ inference
    def forward(self, inputs):
        return self.generator(inputs)

    def predict_step(self, batch, batch_idx=0):
        return self.generator(batch["image"])

    # ------------------------------------------------------------------ optimiser surface (reference parity)
    def configure_optimizers(self):
        """torch Adam + LambdaLR exactly as the reference builds them (used by generic / Lightning callers;
        the fused ``training_step`` applies the same update rule in one HIP launch)."""
        g_optimizer = torch.optim.Adam(self.generator.parameters(), lr=self.lr_g, betas=(0.5, 0.999), eps=1e-7)
        total_iters = self._total_iters()
        g_scheduler = {
            "scheduler": torch.optim.lr_scheduler.LambdaLR(
                g_optimizer, lr_lambda=pix2pix_lr_scheduler(total_iters, 400, total_iters // 2)),
            "interval": "step", "frequency": 1}
        return [g_optimizer], [g_scheduler]

    def _total_iters(self):
        if self.total_iters is not None:
            return int(self.total_iters)
        tr = getattr(self, "_trainer", None) or getattr(self, "trainer", None)
        if tr is not None and hasattr(tr, "estimated_stepping_batches"):
            return int(tr.estimated_stepping_batches)
        raise RuntimeError("set ModelModule.total_iters (or attach a trainer exposing estimated_stepping_batches)")

    def current_lr(self, step=None):
        step = self.global_step_ if step is None else step
        total = self._total_iters()
        return self.lr_g * pix2pix_lr_scheduler(total, 400, total // 2)(step)

    # ------------------------------------------------------------------ fused training step
    def training_step(self, batch, batch_idx=0):
        x, y = batch["image"], batch["target"]
        eng = getattr(self.generator, "_engine", None)
        if eng is None:
            raise NotImplementedError("training_step needs a generator of this package (HIP engine)")
        if not hasattr(eng, "loss_and_grad") or not isinstance(self.loss_reconstruct, WeightedMSELoss):
            return self._training_step_autograd(x, y)
        self.generator.train()
        n8 = batch.get("image_nhwc8")          # written by io_stage.TrainAugmenter next to "image" (same pixels, bf16 NHWC)
        out = eng.forward(x, train=True, img8=n8) if n8 is not None and hasattr(eng, "_decoder_fwd") else eng.forward(x, train=True)
        w = self.loss_reconstruct.marker_weights
        if w.device != out.device:
            self.loss_reconstruct.to(out.device)
            w = self.loss_reconstruct.marker_weights
        sync = self.grad_sync
        # data parallel: the 1/world average rides on dL/d(out) (no pass over the gradient buffer after the exchange)
        gs = sync.begin_step() if sync is not None and hasattr(sync, "begin_step") else 1.0
        loss, dY = eng.loss_and_grad(out, y.to(out.device), w, self.loss_reconstruct.lambda_factor, grad_scale=gs)
        getattr(eng, "backward_fused", eng.backward)(dY, on_decoder_done=(sync.decoder_ready if sync is not None else None),
                                                     on_lora_block_done=(sync.lora_block_done if sync is not None else None))
        if sync is not None:
            sync.finish()
        eng.adam_step(self.current_lr(), betas=(0.5, 0.999), eps=1e-7, max_norm=1.0)
        self.global_step_ += 1
        self.last_loss = loss
        self._nan_guard(loss)
        if self.update_pix_metrics:  # reference models.py:140-143 (the clip to [-0.9, 0.9] is the metrics' own clamp)
            self.train_pix_metrics.update(out, y)
        return loss

    def _training_step_autograd(self, x, y):
        """The reference's own sequence (models.py:87-143) for losses other than WeightedMSELoss: forward and backward through the
        autograd bridge of the HIP engine, torch global-norm clip, torch Adam + LambdaLR."""
        if self.grad_sync is not None:
            raise NotImplementedError("multi-GPU gradient exchange is wired into the fused step (WeightedMSELoss) only")
        if getattr(self, "_opt", None) is None:
            opts, scheds = self.configure_optimizers()
            self._opt, self._sched = opts[0], scheds[0]["scheduler"]
        self.generator.train()
        dev = next(self.generator.parameters()).device
        out = self.generator(x.to(dev))
        y = y.to(out.device)
        if hasattr(self.loss_reconstruct, "marker_weights") and self.loss_reconstruct.marker_weights.device != out.device:
            self.loss_reconstruct.to(out.device)
        loss = self.loss_reconstruct(y_true=y, y_pred=out)
        self._opt.zero_grad()
        loss.backward()
        torch.nn.utils.clip_grad_norm_([p for p in self.generator.parameters() if p.requires_grad], 1.0)
        self._opt.step()
        self._sched.step()
        self.global_step_ += 1
        self.last_loss = loss.detach()
        self._nan_guard(self.last_loss)
        if self.update_pix_metrics:
            self.train_pix_metrics.update(out.detach(), y)
        return self.last_loss

    def on_train_epoch_end(self):
        """reference models.py:207-213: compute, hand to the logger, reset"""
        vals = self.train_pix_metrics.compute()
        self.train_pix_metrics.reset()
        return vals

    def _nan_guard(self, loss=None, flush=False):
        """Host side of the NaN guard (reference models.py:102-105).  The Adam kernel has already refused the update on the
        device if the step was non-finite; here the sticky device flag is copied to pinned memory behind the step (no sync) and
        every copy that has completed is inspected.  ``flush=True`` waits for the outstanding ones (end of training)."""
        eng = getattr(self.generator, "_engine", None)
        flag = eng.nonfinite_flag() if (self.nan_check and eng is not None and hasattr(eng, "nonfinite_flag")) else None
        if flag is not None and not flush:
            host = torch.zeros(1, dtype=torch.int32, pin_memory=True)
            host.copy_(flag, non_blocking=True)
            ev = torch.cuda.Event()
            ev.record()
            self._pending.append((ev, host))
        while self._pending and (flush or self._pending[0][0].query() or len(self._pending) > 8):
            ev, host = self._pending.pop(0)
            ev.synchronize()
            if int(host) != 0:
                self._pending = []
                self._dump_nan_weights(self.checkpoint_state())           # the update was gated: last finite weights
                raise ValueError("Nan found")
        if flag is None and loss is not None and self.nan_check:   # generators without the fused step: the reference's sync check
            if not bool(torch.isfinite(loss)):
                self._dump_nan_weights(self.state_dict())
                raise ValueError("Nan found")

    @staticmethod
    def _dump_nan_weights(state):
        """``weights_nan.ckpt`` (reference models.py:103-104), written by rank 0 only: data-parallel replicas hold identical
        parameters, and every rank writing the same path at once would corrupt the file."""
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized() and dist.get_rank() != 0:
            return
        from .checkpoint import save_checkpoint_atomic
        save_checkpoint_atomic(state, "weights_nan.ckpt")

    def on_train_end(self):
        self._nan_guard(flush=True)

    # ------------------------------------------------------------------ checkpoint / resume
    def checkpoint_state(self):
        """Lightning-shaped checkpoint of the fused trainer: ``state_dict`` with the reference prefixes (``generator.<key>``,
        ``loss_reconstruct.marker_weights``; ``/root/reference/src/inference.py:79-84`` strips them), plus what resuming needs
        and the reference's Lightning checkpoints carry: Adam moments + step count, the LR-schedule position and horizon.
        BatchNorm running statistics are this rank's (data-parallel replicas keep local statistics, as vanilla DDP does)."""
        eng = getattr(self.generator, "_engine", None)
        opt = eng.optimizer_state_dict() if eng is not None and hasattr(eng, "optimizer_state_dict") else None
        if opt is not None and opt.get("exp_avg") is not None:
            opt = dict(opt, exp_avg=opt["exp_avg"].cpu(), exp_avg_sq=opt["exp_avg_sq"].cpu())
        return {"state_dict": {k: v.detach().cpu().clone() for k, v in self.state_dict().items()
                               if k.startswith("generator.") or k.startswith("loss_reconstruct.")},
                "optimizer_state": opt, "global_step": int(self.global_step_), "total_iters": self.total_iters,
                "lr_g": float(self.lr_g), "bn_running_stats": "rank-local"}

    def load_checkpoint_state(self, ckpt):
        sd = {k[len("generator."):]: v for k, v in ckpt["state_dict"].items() if k.startswith("generator.")}
        self.generator.load_state_dict(sd)
        self.global_step_ = int(ckpt.get("global_step", 0))
        # LR horizon: a horizon configured on this module BEFORE loading (run.py sets total_iters = train.max_steps) wins, as
        # Lightning / LambdaLR rebuild the lambda from the new trainer's horizon when a run is resumed with more steps; the
        # checkpoint's value is only adopted when none is configured.  A silent restore of the old horizon would clamp the
        # pix2pix ramp to lr = 0 for every step past it.
        saved = ckpt.get("total_iters")
        if self.total_iters is None:
            self.total_iters = saved
        elif saved is not None and int(saved) != int(self.total_iters):
            import warnings
            warnings.warn(f"resuming at step {self.global_step_} with LR horizon {int(self.total_iters)} steps (the checkpoint "
                          f"was written with {int(saved)}): the pix2pix schedule follows the NEW horizon", stacklevel=2)
        eng = getattr(self.generator, "_engine", None)
        if eng is not None and hasattr(eng, "load_optimizer_state_dict"):
            eng.load_optimizer_state_dict(ckpt.get("optimizer_state"))
        return self

    def _evaluation_step(self, batch, metrics):
        self.generator.eval()
        with torch.no_grad():
            out = self.generator(batch["image"])
            y = batch["target"].to(out.device)
            if self.use_cell_metrics:    # reference evaluation_step, models.py:233-241
                self.cell_metrics.update(out, batch["nuclei"].to(out.device), batch["slide_name"])
            if self.update_pix_metrics:  # reference evaluation_step, models.py:258-261
                metrics.update(out, y)
            return self.loss_reconstruct(y, out)

    def validation_step(self, batch, batch_idx=0):
        return self._evaluation_step(batch, self.val_pix_metrics)

    def test_step(self, batch, batch_idx=0):
        return self._evaluation_step(batch, self.test_pix_metrics)
