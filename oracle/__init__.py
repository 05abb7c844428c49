"""CPU oracle for the MIPHEI-ViT generator hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``oracle/`` is part of the product:
only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` may import it, and only as the checker / the timed CPU baseline.
The product path (``miphei-vit_amd/``) never imports this package and fails
loudly when its HIP library is missing.

What it is: a plain fp32 torch-CPU restatement of the arithmetic the reference
runs for this path (``/root/reference/src/generators/mipheivit.py``,
``lora.py``, ``unet.py:407-438,522-531``, ``loss.py:47-57``,
``utils.py:217-230``, ``models.py:87-143,348-371``) plus the timm-1.0.15
``VisionTransformer`` arithmetic the reference calls but does not contain
(``foundation_models.py:50-57``).

Parity pin: the reference ships no tests or golden vectors for this path
("parity unpinned" by the reference's own tests).  The oracle is pinned instead
against outputs of the reference's own modules imported in the build
container (``oracle/make_golden.py`` -> ``tests/golden/*.npz``) and, for the
timm arithmetic, against Hugging Face ``Dinov2WithRegistersModel``
(transformers 5.15.0) on shared weights.  ``tests/test_oracle_golden.py``
re-checks the oracle against those committed fixtures on every CPU run.
"""

from .detgen import det_normal, det_uniform, det_state_dict  # noqa: F401
from .vit import ViTConfig, vit_forward, VIT_CONFIGS  # noqa: F401
from .decoder import decoder_forward, encoder_regrid  # noqa: F401
from .model import (  # noqa: F401
    generator_forward,
    weighted_mse_loss,
    pix2pix_lr_lambda,
    OracleTrainer,
    orion_marker_weights,
    synth_batch,
)
