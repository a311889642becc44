"""In-process A/B of the round-6 plan switches on the bf16 train-mode forward (same box, same clocks): fuse_bn_finalize, cond_stream.  (Round 6 also tried ONE wait per side stream from the
second stage on - the remaining folds are starved by the first stage kernel and finish with it: +10-14 us at B = 32 x T = 256, reverted.)  tools/exp."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from wavthruvec_pytorch_amd import Generator, synthetic
dev = torch.device('cuda:0')
h = synthetic.make_hparams(num_wv_feat=768)
for B, T in ((32, 256), (64, 512)):
    g = Generator(h)
    g.load_state_dict(synthetic.make_state_dict(h, seed=0))
    g = g.to(dev).train()
    g.precision = 'bf16'
    inp = synthetic.make_inputs(h, B, T, seed=1, device=dev)
    for rep in range(2):
        for bn, cs, mw in ((False, False, False), (True, False, False), (True, False, True), (True, True, True)):
            g.fuse_bn_finalize, g.cond_stream, g.merge_waits = bn, cs, mw
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
            print(f'B={B} T={T} fuse_bn_finalize={bn!s:5} cond_stream={cs!s:5} merge_waits={mw!s:5}: {best * 1e3:.1f} us / forward', flush=True)
