import sys, numpy as np, torch
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
from wavthruvec_pytorch_amd import hipops
from test_hip_ops import _rng, _t, _relayout
dev = torch.device('cuda:0')
for C, L in ((32, 2052), (32, 4096), (64, 2052), (128, 2052)):
    r = _rng(33); B = 3
    x = torch.from_numpy(r.standard_normal((B, C, L), dtype=np.float32)).bfloat16().to(dev)
    a, s_ = _t((1 + 0.2 * r.standard_normal((B, C))).astype(np.float32), dev), _t((0.3 * r.standard_normal((B, C))).astype(np.float32), dev)
    branches = []
    for k in (3, 7, 11):
        ws = [_t(_relayout(torch.from_numpy((r.standard_normal((C, C, k)) / np.sqrt(C * k)).astype(np.float32))).numpy(), dev) for _ in range(2)]
        branches.append(dict(wps1=hipops.pack_split(ws[0], bf16=True), b1=_t(r.standard_normal(C).astype(np.float32) * 0.1, dev),
                             wps2=hipops.pack_split(ws[1], bf16=True), b2=_t(r.standard_normal(C).astype(np.float32) * 0.1, dev), k=k, dil1=1, dil2=3))
    o16 = torch.full((B, C, L), float('nan'), device=dev, dtype=torch.bfloat16)
    o32 = torch.full((B, C, L), float('nan'), device=dev)
    assert hipops.resblock2_stage_split(x, (a, s_), branches, o16, slope=0.1, out_div=3.0, bf16=True, io_bf16=3)
    if C <= 32:
        assert hipops.resblock2_stage_split(x.float(), (a, s_), branches, o32, slope=0.1, out_div=3.0, bf16=True, io_bf16=0)
    else:
        continue
    d = (o16.float() - o32).abs()
    bad = (d > 2.0 ** -7 * o32.abs() + 2e-2)
    print(C, L, 'max', d.max().item(), 'mean', d.mean().item(), 'nbad', int(bad.sum()), 'nan', int(torch.isnan(o16.float()).sum()))
    idx = bad.nonzero()
    print(idx[:10].tolist(), idx[-5:].tolist())
    if len(idx):
        pos = idx[:, 2].cpu().numpy()
        print('positions hist', np.unique(pos // 224, return_counts=True), 'pos%224', np.unique(pos % 224)[:40])
