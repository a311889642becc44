"""MI355X-native Vec2Wav generator forward (the hot path of p1an-lin-jung/WavThruVec_pytorch).

Public surface = the reference's: `Generator`, `ResBlock1`, `ResBlock2`, `ConditionalBatchNorm1d`,
`get_padding`, `init_weights`, `hparams`.  Importing the package does not touch the GPU or load the HIP
library; the first `Generator.forward` does, and raises if the library is missing.
"""
from .models import Generator, ResBlock1, ResBlock2, LRELU_SLOPE  # noqa: F401
from .modules import ConditionalBatchNorm1d  # noqa: F401
from .utils import get_padding, init_weights  # noqa: F401
from . import hparams  # noqa: F401

__all__ = ['Generator', 'ResBlock1', 'ResBlock2', 'ConditionalBatchNorm1d', 'get_padding', 'init_weights',
           'hparams', 'LRELU_SLOPE']
