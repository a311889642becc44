"""bf16 forward in eval mode (running statistics, cached weight fold: what `synthesize` runs) beside the train-mode forward bench.py times.
tools/exp: experiment, not product."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from wavthruvec_pytorch_amd import Generator, synthetic

dev = torch.device('cuda:0')
for B, T in ((32, 256), (64, 512)):
    h = synthetic.make_hparams(num_wv_feat=768)
    g = Generator(h)
    g.load_state_dict(synthetic.make_state_dict(h, seed=0))
    g = g.to(dev)
    g.precision = 'bf16'
    inp = synthetic.make_inputs(h, B, T, seed=1, device=dev)
    for mode in ('train', 'train-fold-cached', 'eval', 'train', 'train-fold-cached', 'eval'):
        g.train(mode != 'eval')
        g.always_refold = mode == 'train'
        with torch.no_grad():
            for _ in range(5):
                g(*inp)
            best = 1e9
            for _ in range(5):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                torch.cuda.synchronize(); e0.record()
                for _ in range(50):
                    g(*inp)
                e1.record(); torch.cuda.synchronize()
                best = min(best, e0.elapsed_time(e1) / 50)
        print(f'B={B} T={T} {mode}: {best * 1e3:.1f} us / forward', flush=True)
