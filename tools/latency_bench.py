#!/usr/bin/env python3
"""Inference latency at small sizes: eager launches vs one captured HIP graph (eval mode, weights folded once)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from wavthruvec_pytorch_amd import Generator, synthetic  # noqa: E402


def main():
    dev = torch.device('cuda:0')
    h = synthetic.make_hparams(num_wv_feat=768)
    g = Generator(h)
    g.load_state_dict(synthetic.make_state_dict(h, seed=0))
    g = g.to(dev).eval()
    g.precision = os.environ.get('V2W_PRECISION', 'f32')
    print('precision', g.precision)
    shapes = [(1, 50), (1, 256), (4, 256)]
    if len(sys.argv) > 2:
        shapes = [(int(sys.argv[1]), int(sys.argv[2]))]
    for B, T in shapes:
        inp = synthetic.make_inputs(h, B, T, seed=1, device=dev)
        with torch.no_grad():
            for _ in range(5):
                g(*inp)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(50):
                g(*inp)
            torch.cuda.synchronize()
            eager = (time.perf_counter() - t0) / 50
            run = g.capture_graph(*inp)
            for _ in range(5):
                run(*inp)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(50):
                run(*inp)
            torch.cuda.synchronize()
            graph = (time.perf_counter() - t0) / 50
        n = B * T * 320
        print(f'B={B} T={T}: eager {eager * 1e3:.3f} ms ({n / eager / 1e6:.1f} M samples/s, RTF {eager / (n / 16000):.5f})   '
              f'graph {graph * 1e3:.3f} ms ({n / graph / 1e6:.1f} M samples/s, RTF {graph / (n / 16000):.5f})')


if __name__ == '__main__':
    main()
