"""MI355X-native (gfx950) implementation of the MIPHEI-ViT generator hot path.

Host side mirrors the reference's Python surface (``get_generator`` / ``ModelModule`` / config keys);
all arithmetic runs in the hand-written HIP kernels of ``csrc/`` through the C-ABI in
``include/miphei_hip.h`` (``libmiphei_hip.so``).  There is no CPU or PyTorch-eager fallback.
"""
__version__ = "0.1.0"
