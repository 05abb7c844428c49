"""Learning-rate schedule of the reference (``src/utils.py:217-230``)."""


def pix2pix_lr_scheduler(total_iters, warmup_iters, decay_start_iter):
    def lr_lambda(step):
        if step < warmup_iters:
            return step / warmup_iters
        if step < decay_start_iter:
            return 1.0
        return max(0.0, 1.0 - (step - decay_start_iter) / (total_iters - decay_start_iter))

    return lr_lambda
