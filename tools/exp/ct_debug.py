import sys, numpy as np, torch, torch.nn.functional as F
sys.path.insert(0, '/root/repo')
from wavthruvec_pytorch_amd import hipops
dev = torch.device('cuda:0')
B, cin, cout, L, k, u = 2, 256, 128, 264, 8, 4
r = np.random.default_rng(23)
x = torch.from_numpy(r.standard_normal((B, cin, L), dtype=np.float32)).bfloat16()
w = (r.standard_normal((cin, cout, k)) / np.sqrt(cin * k / u)).astype(np.float32)
bias = r.standard_normal(cout).astype(np.float32)
xa = F.leaky_relu(x.float(), 0.1).bfloat16().double()
want = F.conv_transpose1d(xa, torch.from_numpy(w).bfloat16().double(), torch.from_numpy(bias).double(), stride=u, padding=(k - u) // 2)
wps = hipops.pack_bf16_convt(torch.from_numpy(np.ascontiguousarray(w.transpose(2, 0, 1))).to(dev), u)
xd = x.to(dev)
out = torch.full((B, cout, L * u), float('nan'), device=dev, dtype=torch.bfloat16)
nt = hipops.convt_bf16_stats_tiles(xd, out, k, u, io_bf16=3)
part = torch.full((nt * cout * 2,), float('nan'), device=dev)
hipops.convt1d_bf16(xd, wps, torch.from_numpy(bias).to(dev), out, k=k, u=u, slope=0.1, stats_part=part, io_bf16=3)
p = part.view(nt, cout, 2).double().cpu()
NT = L * B // nt if False else None
ntl = nt // B
nt_w = 128 if ntl == 3 else 512
print('nt', nt, 'ntl', ntl)
for t in range(nt):
    b, ti = divmod(t, ntl)
    lo, hi = ti * nt_w * u, min((ti + 1) * nt_w, L) * u
    ref = want[b, :, lo:hi].sum(1)
    e = (p[t, :, 0] - ref)
    bad = torch.nonzero(e.abs() > 1e-2).flatten()
    print('tile', t, 'max err', e.abs().max().item(), 'bad channels', bad[:20].tolist(), 'n', len(bad), 'e sample', e[bad[:4]].tolist(), 'ref', ref[bad[:4]].tolist())
