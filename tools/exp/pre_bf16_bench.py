"""conv_pre on the bf16 path (fp32 latents in, bf16 out) at the cfg2 / cfg3 shapes: us per launch, 20 launches after warm-up.
Variants of the tile choice are separate builds selected with V2W_LIB (tools/exp: experiment, not product)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, ctypes as C
from wavthruvec_pytorch_amd import hipops, _hip
dev = 'cuda'
for B, T in ((32, 256), (64, 512), (16, 256)):
    cin = 768 if B != 16 else 1024
    x = torch.randn(B, cin, T, device=dev)
    v = torch.randn(512, cin, 7, device=dev) * 0.02
    frag, sc = torch.empty(hipops.split_halves(7, cin, 512) + 1024, device=dev, dtype=torch.float16), torch.empty(4, device=dev)
    hipops.SplitPlan([(v, None, frag, sc)], torch.device(dev), bf16=True).run()
    wps = (frag, sc)
    out = torch.empty(B, 512, T, device=dev, dtype=torch.bfloat16)
    bias = torch.randn(512, device=dev)
    name = hipops.conv_bf16_config(B, 1, cin, 512, T, 7, 1, 1, io_bf16=2)
    def run():
        hipops.conv1d(x, None, bias, out, k=7, dil=1, slope=1.0, algo=hipops.ALGO_BF16, wps=wps, io_bf16=2)
    ref = None
    for var in ('', '1', '', '1'):                  # V2W_PRE_TILE=1: the 128 x 128 tile (round-6 experiment switch in v2w_conv1d_bf16)
        os.environ.pop('V2W_PRE_TILE', None)
        if var:
            os.environ['V2W_PRE_TILE'] = var
        for _ in range(3):
            run()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); e0.record()
        for _ in range(20):
            run()
        e1.record(); torch.cuda.synchronize()
        t = e0.elapsed_time(e1) / 20 * 1e-3
        o = out.float().clone()
        if ref is None:
            ref = o
        print(f'B={B} T={T} cin={cin} tile-switch={var or 0}: {t * 1e6:.1f} us  {2 * cin * 512 * 7 * B * T / t / 1e12:.0f} TF  max diff vs first {(o - ref).abs().max().item():.2e}')
    os.environ.pop('V2W_PRE_TILE', None)
