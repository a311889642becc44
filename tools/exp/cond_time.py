"""Time of the conditioning chain of a train-mode forward (cond_fc -> cond_sn -> cond_linear) alone.  Round 6: eight rows of W v per pass in cond_sn
(sixteen loads in flight) gave 50.4 -> 47.0 us and different fma contractions (sigma in the 7th digit): not adopted.  tools/exp."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from wavthruvec_pytorch_amd import Generator, synthetic, hipops
dev = torch.device('cuda:0')
h = synthetic.make_hparams(num_wv_feat=768)
g = Generator(h); g.load_state_dict(synthetic.make_state_dict(h, seed=0)); g = g.to(dev).train()
B = 32
x, spk, nz = synthetic.make_inputs(h, B, 8, seed=1, device=dev)
ns = g.num_upsamples
gbs = [torch.empty(B, 2 * c.num_features, device=dev) for c in g.cbns]
z_ws = torch.empty(ns * B * 128, device=dev); sig = torch.empty(ns, device=dev)
def run():
    hipops.cond_gamma_beta(spk, nz, [f.weight.detach() for f in g.fcs], [f.bias.detach() for f in g.fcs], [c.layer.weight_orig.detach() for c in g.cbns],
                           [c.layer.bias.detach() for c in g.cbns], [c.layer.weight_u for c in g.cbns], [c.layer.weight_v for c in g.cbns], gbs, z_ws, sig, True)
for _ in range(5):
    run()
best = 1e9
for _ in range(5):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(20):
        run()
    e1.record(); torch.cuda.synchronize()
    best = min(best, e0.elapsed_time(e1) / 20)
print(f'conditioning chain (3 launches, B = {B}): {best * 1e3:.1f} us   sigma {sig.tolist()}', flush=True)
