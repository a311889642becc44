"""Order of the side stream's folds: the small upsampler folds in front of the Conv1d batch (one event then stands for all of them) against behind it.
Needs the experiment's env switch V2W_UPS_FIRST in Generator._split_weights (adopted since: upsamplers first; then the same A/B for the events between the upsampler folds - V2W_FEW_MARKS, also adopted: -14.3 / +1.8 / -1.9 us; and V2W_REST_EARLY - the Conv1d batch in front of ups.1 .. 4 with the one event behind all of them - not adopted: +12.2 / -7.1 / +1.4 us;
measured -8.4 / -2.6 / -6.2 us at B x T = 32 x 256 / 64 x 512 / 16 x 256).  tools/exp."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from wavthruvec_pytorch_amd import Generator, synthetic
dev = torch.device('cuda:0')
h = synthetic.make_hparams(num_wv_feat=768)
for B, T in ((32, 256), (64, 512), (16, 256)):
    inp = synthetic.make_inputs(h, B, T, seed=1, device=dev)
    res = {}
    for rep in range(3):
        for uf in ('', '1'):
            os.environ.pop('V2W_UPS_FIRST', None)
            if uf:
                os.environ['V2W_UPS_FIRST'] = '1'
            g = Generator(h)
            g.load_state_dict(synthetic.make_state_dict(h, seed=0))
            g = g.to(dev).train()
            g.precision = 'bf16'
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
            res[uf] = min(res.get(uf, 1e9), best)
            del g
    print(f'B={B} T={T}: rest first {res[""] * 1e3:.1f} us   upsamplers first {res["1"] * 1e3:.1f} us  ({(res["1"] - res[""]) * 1e3:+.1f})', flush=True)
