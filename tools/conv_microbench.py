#!/usr/bin/env python3
"""Run single conv launches of the generator's shapes in a loop (for rocprofv3 --pmc / timing of one kernel)."""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

SHAPES = {   # name: (B, Cin, Cout, L, k, dil, u)
    's0k3': (32, 256, 256, 1280, 3, 1, 1), 's0k11': (32, 256, 256, 1280, 11, 3, 1),
    's1k3': (32, 128, 128, 5120, 3, 1, 1), 's1k7': (32, 128, 128, 5120, 7, 3, 1), 's1k11': (32, 128, 128, 5120, 11, 3, 1),
    's2k3': (32, 64, 64, 20480, 3, 1, 1), 's2k11': (32, 64, 64, 20480, 11, 3, 1),
    's3k3': (32, 32, 32, 40960, 3, 1, 1), 's3k11': (32, 32, 32, 40960, 11, 3, 1),
    's4k3': (32, 16, 16, 81920, 3, 1, 1), 's4k11': (32, 16, 16, 81920, 11, 3, 1),
    'pre': (32, 768, 512, 256, 7, 1, 1), 'up0': (32, 512, 256, 256, 11, 1, 5), 'up2': (32, 128, 64, 5120, 8, 1, 4),
}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('shapes', nargs='*', default=list(SHAPES))
    ap.add_argument('--reps', type=int, default=5)
    ap.add_argument('--flags', type=int, default=1, help='1: residual + affine + accumulate (conv2-like epilogue)')
    args = ap.parse_args()
    from wavthruvec_pytorch_amd import hipops
    dev = torch.device('cuda:0')
    for name in args.shapes:
        B, ci, co, L, k, d, u = SHAPES[name]
        x = torch.randn(B, ci, L, device=dev)
        w = torch.randn(k, ci, co, device=dev) / (ci * k) ** 0.5
        wp = hipops.pack_mfma(w, u=u)
        bias = torch.randn(co, device=dev)
        out = torch.zeros(B, co, L * u, device=dev)
        a = torch.rand(B, ci, device=dev) + 0.5
        s = torch.randn(B, ci, device=dev) * 0.1
        flops = 2.0 * ci * co * k * L * B
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(args.reps + 1)]
        for r in range(args.reps + 1):
            if u == 1:
                if args.flags and ci == co:
                    hipops.conv1d(x, None, bias, out, k=k, dil=d, slope=0.1, in_affine=(a, s), res=x, res_affine=(a, s),
                                  accumulate=True, wp=wp)
                else:
                    hipops.conv1d(x, None, bias, out, k=k, dil=d, slope=0.1, wp=wp)
            else:
                hipops.convt1d(x, None, bias, out, k=k, u=u, slope=0.1, wp=wp)
            ev[r].record()
        torch.cuda.synchronize()
        ts = sorted(ev[i].elapsed_time(ev[i + 1]) for i in range(args.reps))
        med = ts[len(ts) // 2] * 1e-3
        print(f'{name:6s} {hipops.conv_tile_config(B, ci, co, L, k, d, u)}  {med * 1e6:8.1f} us  {flops / med / 1e12:6.1f} TF')


if __name__ == '__main__':
    main()
