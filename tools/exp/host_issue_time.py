"""Host time to ISSUE a bf16 forward (launch-plan replay) beside the GPU time of the forward: is the timed loop host-bound?  tools/exp."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from wavthruvec_pytorch_amd import Generator, synthetic
dev = torch.device('cuda:0')
h = synthetic.make_hparams(num_wv_feat=768)
for B, T in ((32, 256), (1, 50)):
    g = Generator(h)
    g.load_state_dict(synthetic.make_state_dict(h, seed=0))
    g = g.to(dev)
    g.precision = 'bf16'
    inp = synthetic.make_inputs(h, B, T, seed=1, device=dev)
    for mode in ('train', 'eval'):
        g.train(mode == 'train')
        with torch.no_grad():
            for _ in range(5):
                g(*inp)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(50):
                g(*inp)
            t1 = time.perf_counter()
            torch.cuda.synchronize()
            t2 = time.perf_counter()
        print(f'B={B} T={T} {mode}: host issue {(t1 - t0) / 50 * 1e6:.0f} us / forward, wall {(t2 - t0) / 50 * 1e6:.0f} us / forward', flush=True)
