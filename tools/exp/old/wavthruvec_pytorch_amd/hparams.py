"""Generator-side hyper-parameters with the attribute names of the reference's vec2wav/hparams.py
(lines 25-27, 30, 40-44, 51).  `Generator(h)` accepts this module, the reference's own hparams module,
or any attribute bag with these nine names."""

# vec2wav
n_feat_dim = 1024  # wav2vec 2.0 feature dim
spk_dim = 192
noise_dim = 192

# hifi-gan: NOTE an *int* - `h.resblock == '1'` is False, so ResBlock2 is built
resblock = 1

# generator
upsample_rates = [5, 4, 4, 2, 2]
upsample_kernel_sizes = [11, 8, 8, 4, 4]
upsample_initial_channel = 512
resblock_kernel_sizes = [3, 7, 11]
resblock_dilation_sizes = [[1, 3, 5], [1, 3, 5], [1, 3, 5]]

num_wv_feat = 1024
sampling_rate = 16000

dist_config = {
    "dist_backend": "nccl",  # = RCCL on PyTorch-ROCm
    "dist_url": "tcp://127.0.0.1:54321",
    "world_size": 1,
}
