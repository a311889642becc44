"""What the cfg2 bf16 forward would take without (a) the BatchNorm reduce / finalize launches between the stages, (b) conv_pre's weight fold on
the main stream - an upper bound on what fusing them away can return (results are garbage in the what-if runs; only the clock matters).
tools/exp: experiment, not product."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from wavthruvec_pytorch_amd import Generator, synthetic, hipops

B, T = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (32, 256)
dev = torch.device('cuda:0')
real = {n: getattr(hipops, n) for n in ('bn_reduce_partials', 'bn_finalize', 'bn_reduce_finalize_slices')}


REAL_WS = {}


def measure(tag, skip_bn=False):
    for n, f in real.items():
        setattr(hipops, n, (lambda *a, **k: None) if skip_bn else f)
    h = synthetic.make_hparams(num_wv_feat=768)
    g = Generator(h)
    g.load_state_dict(synthetic.make_state_dict(h, seed=0))
    g = g.to(dev).train()
    g.precision = 'bf16'
    g.always_refold = True
    inp = synthetic.make_inputs(h, B, T, seed=1, device=dev)
    with torch.no_grad():
        for _ in range(5):
            g(*inp)
        if skip_bn:            # the affines of a real forward on the same inputs: the what-if forwards compute on real values (same clocks)
            for k, v in REAL_WS.items():
                g._ws[k].copy_(v)
            for _ in range(3):
                g(*inp)
        else:
            REAL_WS.update({k: v.clone() for k, v in g._ws.items() if k.startswith(('bn.a', 'bn.s')) and not k.startswith(('bn.stats', 'bn.slices'))})
        best = 1e9
        for _ in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize(); e0.record()
            for _ in range(50):
                g(*inp)
            e1.record(); torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) / 50)
    print(f'{tag}: {best * 1e3:.1f} us / forward (B={B}, T={T})', flush=True)


measure('as shipped')
measure('no BN reduce / finalize launches', skip_bn=True)
measure('as shipped')
