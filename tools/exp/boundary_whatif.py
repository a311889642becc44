"""What the cfg2 bf16 train-mode forward would take without (a) the BatchNorm reduce / finalize launches between the stages, (b) the main stream's
waits for side-stream events, (c) the weight folds - upper bounds on what removing each can return.  The what-if forwards compute on real values
(the affines of a real forward; the folds rewrite the same weights), only the clock matters.  tools/exp: experiment, not product."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from wavthruvec_pytorch_amd import Generator, synthetic, hipops, schedule

B, T = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (32, 256)
dev = torch.device('cuda:0')
real = {n: getattr(hipops, n) for n in ('bn_reduce_partials', 'bn_finalize', 'bn_reduce_finalize_slices')}
real_need = schedule.Recorder.need
REAL_WS = {}


def measure(tag, skip_bn=False, skip_need=False, refold=True, mode='train'):
    for n, f in real.items():
        setattr(hipops, n, (lambda *a, **k: None) if skip_bn else f)
    schedule.Recorder.need = (lambda self, name: None) if skip_need else real_need
    h = synthetic.make_hparams(num_wv_feat=768)
    g = Generator(h)
    g.load_state_dict(synthetic.make_state_dict(h, seed=0))
    g = g.to(dev).train(mode == 'train')
    g.precision = 'bf16'
    g.always_refold = refold
    inp = synthetic.make_inputs(h, B, T, seed=1, device=dev)
    with torch.no_grad():
        for _ in range(5):
            g(*inp)
        if skip_bn:            # the affines of a real forward on the same inputs
            for k, v in REAL_WS.items():
                g._ws[k].copy_(v)
            for _ in range(3):
                g(*inp)
        elif mode == 'train':
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
measure('no waits for side-stream events', skip_need=True)
measure('neither', skip_bn=True, skip_need=True)
measure('fold cached', refold=False)
measure('fold cached, no BN launches, no waits', skip_bn=True, skip_need=True, refold=False)
measure('eval mode', mode='eval')
measure('as shipped')
