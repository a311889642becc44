import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
from wavthruvec_pytorch_amd import Generator, synthetic
dev = torch.device('cuda:0')
for (B, T) in ((64, 512), (32, 256)):
    h = synthetic.make_hparams(num_wv_feat=768)
    g = Generator(h); g.load_state_dict(synthetic.make_state_dict(h, seed=0)); g = g.to(dev).train(); g.precision = 'bf16'
    inp = synthetic.make_inputs(h, B, T, seed=4, device=dev)
    el, ms = bench.run_steps(g, inp, 40, 8)
    print(f'B={B} T={T} eager: wall {el / 40 * 1e3:.3f} ms/step, event median {sorted(ms)[20]:.3f}')
    run = g.capture_graph(*inp)
    y0 = run(*inp).clone()
    with torch.no_grad():
        y1 = g(*inp)
    print('  graph vs eager max|dy|', (y0 - y1).abs().max().item())
    for _ in range(8): run(*inp)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    evs = [torch.cuda.Event(enable_timing=True) for _ in range(41)]
    evs[0].record()
    for i in range(40):
        run(*inp); evs[i + 1].record()
    torch.cuda.synchronize(); el = time.perf_counter() - t0
    ms = sorted(evs[i].elapsed_time(evs[i + 1]) for i in range(40))
    print(f'  graph replay: wall {el / 40 * 1e3:.3f} ms/step, event median {ms[20]:.3f}')
