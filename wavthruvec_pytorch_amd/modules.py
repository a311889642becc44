"""`ConditionalBatchNorm1d` with the reference's surface (vec2wav/modules.py:5-30): same constructor,
same `forward(inputs, noise)`, same `state_dict` keys - including the reference's `batch_nrom` typo -
with the arithmetic in HIP kernels (wavthruvec_pytorch_amd/csrc/v2w_cbn.hip)."""
from __future__ import annotations

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import _hip, hipops


class _BatchNormState(nn.Module):
    """Buffers of `nn.BatchNorm1d(num_features, affine=False)` (modules.py:14); no arithmetic here."""

    def __init__(self, num_features, eps=1e-5, momentum=0.1):
        super().__init__()
        self.num_features = num_features
        self.eps = eps
        self.momentum = momentum
        self.affine = False
        self.track_running_stats = True
        self.register_buffer('running_mean', torch.zeros(num_features))
        self.register_buffer('running_var', torch.ones(num_features))
        self.register_buffer('num_batches_tracked', torch.tensor(0, dtype=torch.long))

    def extra_repr(self):
        return f'{self.num_features}, eps={self.eps}, momentum={self.momentum}, affine=False'


class _SpectralNormLinearState(nn.Module):
    """Parameters/buffers of legacy `spectral_norm(nn.Linear(z, 2C))` (modules.py:16-18):
    `bias`, `weight_orig` (params), `weight_u`, `weight_v` (buffers).  The power iteration runs in the
    `cond_sn_kernel` HIP kernel and updates u/v in place in train mode, as the reference's hook does."""

    def __init__(self, in_features, out_features):
        super().__init__()
        self.in_features, self.out_features = in_features, out_features
        self.bias = nn.Parameter(torch.zeros(out_features))                                  # modules.py:18
        self.weight_orig = nn.Parameter(torch.empty(out_features, in_features).normal_(1, 0.02))  # modules.py:17
        self.register_buffer('weight_u', F.normalize(torch.randn(out_features), dim=0, eps=1e-12))
        self.register_buffer('weight_v', F.normalize(torch.randn(in_features), dim=0, eps=1e-12))

    def extra_repr(self):
        return f'in_features={self.in_features}, out_features={self.out_features}, spectral_norm'


class ConditionalBatchNorm1d(nn.Module):
    """Conditional Batch Normalization (reference: vec2wav/modules.py:5-30)."""

    def __init__(self, num_features, z_channels=128):
        super().__init__()
        if z_channels != 128:
            raise ValueError('the HIP conditioning kernels are built for z_channels == 128 (models.py:110, Z_CHANNEL)')
        self.num_features = num_features
        self.z_channels = z_channels
        self.batch_nrom = _BatchNormState(num_features)           # sic: the reference's attribute name
        self.layer = _SpectralNormLinearState(z_channels, num_features * 2)

    @_hip.on_tensor_device
    def forward(self, inputs, noise):
        """outputs = gamma(noise) * BN(inputs) + beta(noise); materialises the result (standalone use).

        `Generator.forward` does not call this: it folds the affine into its consumers' loads."""
        if not inputs.is_cuda:
            raise RuntimeError('ConditionalBatchNorm1d: the HIP path needs GPU tensors; there is no CPU fallback')
        x = inputs.contiguous().float()
        z = noise.contiguous().float()
        B, C, L = x.shape
        dev = x.device
        gb = torch.empty((B, 2 * C), device=dev)
        z_ws = torch.empty((B, 128), device=dev)
        sigma = torch.empty((1,), device=dev)
        a = torch.empty((B, C), device=dev)
        s = torch.empty((B, C), device=dev)
        bn, ly = self.batch_nrom, self.layer
        stats = None
        if self.training:
            stats = torch.empty((2 * C + 1,), device=dev, dtype=torch.float64)
            part = torch.empty((2 * C * 64,), device=dev, dtype=torch.float64)
            hipops.bn_stats(x, stats, part)
        # BN first, then the spectral-norm hook inside layer() (modules.py:23-24) - same mutation order
        _cond_no_fc(z, ly, gb, z_ws, sigma, self.training)
        hipops.bn_finalize(stats, gb, bn.running_mean, bn.running_var, bn.num_batches_tracked, a, s,
                           training=self.training, momentum=bn.momentum, eps=bn.eps)
        return hipops.affine_apply(x, a, s, torch.empty_like(x))


def _cond_no_fc(z, ly, gb, z_ws, sigma, training):
    import ctypes as C
    from . import _hip
    args = _hip.CondArgs()
    args.spk = z.data_ptr(); args.noise = None
    args.sn_w[0] = ly.weight_orig.data_ptr(); args.sn_b[0] = ly.bias.data_ptr()
    args.sn_u[0] = ly.weight_u.data_ptr(); args.sn_v[0] = ly.weight_v.data_ptr()
    args.gb[0] = gb.data_ptr(); args.C[0] = ly.out_features // 2
    args.z_ws = z_ws.data_ptr(); args.sigma_ws = sigma.data_ptr()
    args.n_stages = 1; args.B = z.shape[0]; args.spk_dim = z.shape[1]; args.noise_dim = 0
    args.training = int(training)
    _hip.check(_hip.load().v2w_cond_gamma_beta(C.byref(args), torch.cuda.current_stream(z.device).cuda_stream),
               'v2w_cond_gamma_beta')
