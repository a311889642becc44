"""In-process A/B of experiment bits read from V2W_EXP at plan time (a fresh Generator per setting) on the bf16 train-mode forward.  Round 6 tried
through it (none adopted, all inside +-6 us of noise or worse): one fork event for both side streams instead of one per stream (1 298 against 1 296 us), the
two-level statistics reduce only from 4 096 rows with four row loads in flight in the one-level kernel (-5 us), the waits for the conditioning chain and the Conv1d
fragments issued together in front of the first statistics launch (+9 us).  tools/exp."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from wavthruvec_pytorch_amd import Generator, synthetic
dev = torch.device('cuda:0')
h = synthetic.make_hparams(num_wv_feat=768)
bits = [int(b) for b in (sys.argv[1].split(',') if len(sys.argv) > 1 else '0,2,4,6'.split(','))]
for B, T in ((32, 256), (64, 512), (16, 256)):
    inp = synthetic.make_inputs(h, B, T, seed=1, device=dev)
    res = {}
    for rep in range(3):
        for e in bits:
            os.environ['V2W_EXP'] = str(e)
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
            res[e] = min(res.get(e, 1e9), best)
            del g
    print(f'B={B} T={T}: ' + '   '.join(f'V2W_EXP={e}: {res[e] * 1e3:.1f} us' for e in bits), flush=True)
