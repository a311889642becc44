"""N bf16 forwards in one mode at one shape, nothing else (for rocprofv3 --kernel-trace + tools/trace_gaps.py): argv B T mode n.  tools/exp."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from wavthruvec_pytorch_amd import Generator, synthetic
B, T, mode, n = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], int(sys.argv[4])
dev = torch.device('cuda:0')
h = synthetic.make_hparams(num_wv_feat=768)
g = Generator(h)
g.load_state_dict(synthetic.make_state_dict(h, seed=0))
g = g.to(dev).train(mode != 'eval')
g.precision = 'bf16'
inp = synthetic.make_inputs(h, B, T, seed=1, device=dev)
with torch.no_grad():
    for _ in range(n):
        g(*inp)
torch.cuda.synchronize()
