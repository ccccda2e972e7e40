"""sepkern -- host side of the MI355X-native uPIT hot path.

    _lib     ctypes binding of libsepkern.so (the C ABI in include/sepkern.h); no fallback
    ops      tensor-level wrappers (GEMM, STFT/iSTFT, BLSTM recurrence, BN, PIT-MSE, clip+Adam)
    engine   the network's forward/backward as a sequence of those calls over flat param/grad buffers
    optim    fused clip_grad_norm_ + Adam over the flat buffers
    synth    WSJ0-2mix-shaped synthetic data (wav trees, id lists, HBM-resident batches)
    sisdr    SI-SDR scoring
"""
from ._lib import SepkernError, load  # noqa: F401
