#!/usr/bin/env python3
"""vec2wav inference entry: text2vec latents + speaker embedding -> 16 kHz waveform on the MI355X HIP path.

The reference has no such script: `text2vec/eval.py:121-122` writes `*_feat_postnet.npy` of shape
(1, T, n_feat_dim) and the only Generator inference code is the validation loop of `vec2wav/train.py:246-291`.
This closes the text2vec -> vec2wav hand-off with the reference's own wire formats:

  generator checkpoint   `g_%08d` = torch.save({'generator': state_dict})          vec2wav/train.py:228-230, utils.py:39-58
  latents                `.npy` (1, T, C) or (T, C) float32                        text2vec/eval.py:121-122, prepare_data.py
  speaker embedding      `{spk}.pth` tensor (1, 1, 192) -> squeezed to (1, 192)    vec2wav/pre_spk_emb.py, dataset.py:181-185
  noise                  randn(1, noise_dim)                                       vec2wav/train.py:256

    python -m wavthruvec_pytorch_amd.synthesize --checkpoint run/g_00100000 --feat a_feat_postnet.npy \
        --spk-emb SSB0005.pth --out a.wav [--num-wv-feat 768] [--remove-weight-norm] [--seed 1234]
"""
from __future__ import annotations

import argparse
import os
import wave

import numpy as np
import torch

from . import synthetic
from .models import Generator
from .utils import load_checkpoint, scan_checkpoint


def load_latents(path: str) -> torch.Tensor:
    """`.npy` (1, T, C) / (T, C) -> channels-first (1, C, T) float32 (the permute of dataset.py:212-213)."""
    a = np.load(path)
    if a.ndim == 2:
        a = a[None]
    if a.ndim != 3 or a.shape[0] != 1:
        raise ValueError(f'{path}: expected latents of shape (1, T, C) or (T, C), got {a.shape}')
    return torch.from_numpy(np.ascontiguousarray(a.astype(np.float32))).permute(0, 2, 1).contiguous()


def load_speaker_embedding(path: str) -> torch.Tensor:
    """`{spk}.pth` saved as (1, 1, 192) (pre_spk_emb.py) -> (1, 192) float32."""
    t = torch.load(path, map_location='cpu')
    t = torch.as_tensor(t, dtype=torch.float32)
    return t.reshape(1, -1).contiguous()


def write_wav(path: str, audio: torch.Tensor, sampling_rate: int) -> None:
    """(1, 1, N) float in [-1, 1] -> 16-bit PCM mono wav (what train.py's SummaryWriter.add_audio consumers expect)."""
    pcm = (audio.detach().reshape(-1).clamp(-1.0, 1.0).cpu().numpy() * 32767.0).round().astype('<i2')
    with wave.open(path, 'wb') as w:
        w.setnchannels(1)
        w.setsampwidth(2)
        w.setframerate(sampling_rate)
        w.writeframes(pcm.tobytes())


def build_generator(checkpoint: str, h, device, remove_weight_norm: bool = False) -> Generator:
    """`checkpoint` is a `g_%08d` file or a directory holding them (newest is taken, utils.py:53-58)."""
    path = checkpoint
    if os.path.isdir(checkpoint):
        path = scan_checkpoint(checkpoint, 'g_')
        if path is None:
            raise FileNotFoundError(f'no g_???????? checkpoint in {checkpoint}')
    sd = load_checkpoint(path, 'cpu')['generator']
    g = Generator(h)
    g.load_state_dict(sd)
    g = g.to(device).eval()
    if remove_weight_norm:
        g.remove_weight_norm()      # folded weights are cached in eval mode either way; this mirrors the HiFi-GAN idiom
    return g


@torch.no_grad()
def synthesize(g: Generator, feat: torch.Tensor, spk_emb: torch.Tensor, seed: int = 1234, noise=None) -> torch.Tensor:
    dev = next(g.parameters()).device
    if noise is None:
        gen = torch.Generator(device='cpu').manual_seed(seed)
        noise = torch.randn(feat.shape[0], g.h.noise_dim, generator=gen)
    return g(feat.to(dev), spk_emb.to(dev), noise.to(dev))


def main(argv=None) -> int:
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument('--checkpoint', required=True, help='g_%%08d file or directory')
    ap.add_argument('--feat', required=True, help='text2vec *_feat_postnet.npy (1, T, C)')
    ap.add_argument('--spk-emb', required=True, help='{spk}.pth (1, 1, 192)')
    ap.add_argument('--out', required=True, help='output .wav')
    ap.add_argument('--num-wv-feat', type=int, default=None, help='latent width (default: taken from the .npy)')
    ap.add_argument('--resblock', default=1, help="'1' selects ResBlock1 (string!), anything else ResBlock2 (reference default)")
    ap.add_argument('--sampling-rate', type=int, default=16000)
    ap.add_argument('--remove-weight-norm', action='store_true')
    ap.add_argument('--seed', type=int, default=1234)
    ap.add_argument('--device', default='cuda:0')
    ap.add_argument('--precision', default='f32', choices=['f32', 'f16x3', 'bf16'],
                    help="Generator.precision: exact fp32 (default), split-f16 (fp32-level accuracy, faster) or bf16 operands")
    args = ap.parse_args(argv)
    feat = load_latents(args.feat)
    spk = load_speaker_embedding(args.spk_emb)
    resblock = '1' if str(args.resblock) == "'1'" or args.resblock == '1s' else args.resblock
    h = synthetic.make_hparams(num_wv_feat=args.num_wv_feat or feat.shape[1], resblock=resblock)
    g = build_generator(args.checkpoint, h, torch.device(args.device), args.remove_weight_norm)
    g.precision = args.precision
    y = synthesize(g, feat, spk, seed=args.seed)
    write_wav(args.out, y, args.sampling_rate)
    print(f'{args.out}: {y.shape[-1]} samples ({y.shape[-1] / args.sampling_rate:.2f} s) from {feat.shape[-1]} frames')
    return 0


if __name__ == '__main__':
    raise SystemExit(main())
