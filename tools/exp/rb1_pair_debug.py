import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, torch.nn.functional as F
from wavthruvec_pytorch_amd import hipops
dev = torch.device('cuda:0')
B, C, L = 3, 16, 4100
for dil in (1, 3, 5):
    g = torch.Generator().manual_seed(500 + C + L + dil)
    ks = [3, 7, 11]
    xs = [torch.randn(B, C, L, generator=g).bfloat16() for _ in ks]
    a = 1 + 0.2 * torch.randn(B, C, generator=g)
    s_ = 0.2 * torch.randn(B, C, generator=g)
    w1 = [torch.randn(C, C, k, generator=g) / (C * k) ** 0.5 for k in ks]
    w2 = [torch.randn(C, C, k, generator=g) / (C * k) ** 0.5 for k in ks]
    b1 = [0.1 * torch.randn(C, generator=g) for _ in ks]
    b2 = [0.1 * torch.randn(C, generator=g) for _ in ks]
    def ref(x, j):
        xa = (a[:, :, None] * x.float() + s_[:, :, None])
        xact = F.leaky_relu(xa, 0.1).bfloat16().double()
        u = F.conv1d(xact, w1[j].bfloat16().double(), b1[j].double(), dilation=dil, padding=dil * (ks[j] - 1) // 2)
        uact = F.leaky_relu(u.float(), 0.1).bfloat16().double()
        return xa.double() + F.conv1d(uact, w2[j].bfloat16().double(), b2[j].double(), padding=(ks[j] - 1) // 2)
    br = [dict(wps1=hipops.pack_split(w1[j].permute(2, 1, 0).contiguous().to(dev), bf16=True), b1=b1[j].to(dev),
               wps2=hipops.pack_split(w2[j].permute(2, 1, 0).contiguous().to(dev), bf16=True), b2=b2[j].to(dev), k=ks[j], dil1=dil, dil2=1) for j in range(3)]
    xd = [x.to(dev) for x in xs]
    outs = [torch.full((B, C, L), float('nan'), device=dev, dtype=torch.bfloat16) for _ in ks]
    assert hipops.resblock1_pairs_bf16(xd, (a.to(dev), s_.to(dev)), br, outs, slope=0.1)
    for j in range(3):
        want = ref(xs[j], j)
        err = (outs[j].cpu().double() - want).abs()
        bad = (err > 2.0 ** -8 * want.abs() + 2e-2)
        idx = bad.nonzero()
        print(f'dil {dil} k {ks[j]}: max err {err.max().item():.4f} nan {torch.isnan(outs[j].float()).sum().item()} bad {bad.sum().item()}',
              'first bad (b,c,l):', idx[:3].tolist(), 'bad positions range', (idx[:, 2].min().item(), idx[:, 2].max().item()) if len(idx) else None)
