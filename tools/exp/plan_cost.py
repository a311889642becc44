"""Host cost of the first forward of a new shape (plan + record + tape finalize) against a replay.  tools/exp."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from wavthruvec_pytorch_amd import Generator, synthetic, schedule
dev = torch.device('cuda:0')
h = synthetic.make_hparams(num_wv_feat=768)
g = Generator(h)
g.load_state_dict(synthetic.make_state_dict(h, seed=0))
g = g.to(dev).eval()
fin = schedule.Tape.finalize
spent = [0.0]
def timed_finalize(self, binds, owned=None):
    t = time.perf_counter(); r = fin(self, binds, owned); spent[0] += time.perf_counter() - t; return r
schedule.Tape.finalize = timed_finalize
own = Generator._owned_ranges
spent_own = [0.0]
def timed_own(self):
    t = time.perf_counter(); r = own(self); spent_own[0] += time.perf_counter() - t; return r
Generator._owned_ranges = timed_own
for prec in ('f32', 'bf16'):
    g.precision = prec
    with torch.no_grad():
        g(*synthetic.make_inputs(h, 1, 40, seed=1, device=dev)); torch.cuda.synchronize()
        for T in (52, 100, 256):
            inp = synthetic.make_inputs(h, 1, T, seed=1, device=dev)
            ts = []
            for _ in range(3):
                spent[0] = spent_own[0] = 0.0
                torch.cuda.synchronize(); t0 = time.perf_counter(); g(*inp); torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
                if _ == 0:
                    first = (spent[0] * 1e3, spent_own[0] * 1e3)
            print(f'{prec} B=1 T={T}: first forward {ts[0]:.1f} ms (finalize {first[0]:.1f} ms, owned_ranges {first[1]:.1f} ms), then {ts[1]:.2f}, {ts[2]:.2f} ms', flush=True)
