"""Side streams at normal vs high priority (their kernels are small and latency-bound; beside a saturating stage kernel they are starved):
in-process A/B on the bf16 train-mode forward and the exact-fp32 one.  Needs `_SIDE_PRIORITY` in models.py (the experiment's one-line patch:
`torch.cuda.Stream(device=device, priority=_SIDE_PRIORITY)` in Generator._side_stream); measured round 6: +4 / -10 / -2 / -11 us on forwards of
1 441 / 4 869 / 883 / 8 655 us - inside the noise, not adopted.  tools/exp."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from wavthruvec_pytorch_amd import Generator, synthetic, models
dev = torch.device('cuda:0')
h = synthetic.make_hparams(num_wv_feat=768)
for prec, B, T in (('bf16', 32, 256), ('bf16', 64, 512), ('bf16', 16, 256), ('f32', 32, 256)):
    inp = synthetic.make_inputs(h, B, T, seed=1, device=dev)
    res = {}
    for rep in range(2):
        for pr in (0, -1):
            models._SIDE_PRIORITY = pr
            models._SIDE_STREAMS.clear()
            g = Generator(h)
            g.load_state_dict(synthetic.make_state_dict(h, seed=0))
            g = g.to(dev).train()
            g.precision = prec
            n = 40 if prec == 'bf16' else 10
            with torch.no_grad():
                for _ in range(4):
                    g(*inp)
                best = 1e9
                for _ in range(4):
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    torch.cuda.synchronize(); e0.record()
                    for _ in range(n):
                        g(*inp)
                    e1.record(); torch.cuda.synchronize()
                    best = min(best, e0.elapsed_time(e1) / n)
            res[pr] = min(res.get(pr, 1e9), best)
            del g
            torch.cuda.empty_cache()
    print(f'{prec} B={B} T={T}: side priority 0: {res[0] * 1e3:.1f} us   -1 (high): {res[-1] * 1e3:.1f} us  ({(res[-1] - res[0]) * 1e3:+.1f})', flush=True)
