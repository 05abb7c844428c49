"""Learning-rate schedule used by ``ModelModule`` (same curve as the reference's ``pix2pix_lr_scheduler``,
``/root/reference/src/utils.py:217-230``): linear warm-up over ``warmup_iters`` steps, plateau at 1 until
``decay_start_iter``, then a linear ramp that reaches 0 at ``total_iters``."""


def pix2pix_lr_scheduler(total_iters, warmup_iters, decay_start_iter):
    span = float(total_iters - decay_start_iter)

    def lr_lambda(step):
        if step < warmup_iters:
            return step / warmup_iters          # 0 at step 0: the first optimiser step is a no-op, as in the reference
        ramp = 1.0 - (step - decay_start_iter) / span if step >= decay_start_iter else 1.0
        return ramp if ramp > 0.0 else 0.0

    return lr_lambda
