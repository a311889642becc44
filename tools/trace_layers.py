#!/usr/bin/env python3
"""Per-layer view of a rocprofv3 --kernel-trace CSV of bench.py: duration, TFLOP/s and algorithmic GB/s per conv launch."""
import csv
import glob
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from wavthruvec_pytorch_amd import workmodel, synthetic  # noqa: E402


def main(path, B=32, T=256, which=-2, act_bytes=4):
    f = glob.glob(os.path.join(path, '**', '*_kernel_trace.csv'), recursive=True)[0]
    rows = list(csv.DictReader(open(f)))
    rows.sort(key=lambda r: int(r['Start_Timestamp']))
    names = [r['Kernel_Name'] for r in rows]
    idx = [i for i, n in enumerate(names) if 'cond_fc' in n]
    s, e = idx[which - 1], idx[which]
    h = synthetic.make_hparams(num_wv_feat=768)
    by_name = {l['name']: l for l in workmodel.conv_layers(h, B, T, act_bytes)}
    # launch groups as Generator.forward issues them: the residual branches of a stage share launches (heaviest first)
    groups = [['conv_pre']]
    nk = len(h.resblock_kernel_sizes)
    ns = len(h.upsample_rates)
    for i in range(ns):
        js = sorted(range(nk), key=lambda j: -h.resblock_kernel_sizes[j])
        if not (act_bytes == 2 and i > 0):    # bf16 tensors (round 4): ups[i], i > 0, runs inside the kernel of stage i - 1 (Generator.fuse_up)
            groups.append([f'ups.{i}'])
        C = h.upsample_initial_channel // 2 ** (i + 1)
        if C in (16, 32) or act_bytes == 2:   # Generator.fuse_stage default (and every stage on bf16 tensors): the whole residual section is ONE launch
            g = [f'resblocks.{i * nk + j}.{c}' for j in range(nk) for c in (0, 1)]
            if act_bytes == 2:
                g += [f'ups.{i + 1}'] if i + 1 < ns else ['conv_post']      # ... with the next upsampler / the generator's tail behind it
            groups.append(g)
        else:
            groups.append([f'resblocks.{i * nk + j}.0' for j in js])
            groups.append([f'resblocks.{i * nk + j}.1' for j in js if j < nk - 1])
            groups.append([f'resblocks.{i * nk + nk - 1}.1'])
    if act_bytes != 2:
        groups.append(['conv_post'])
    li = 0
    tot = 0.0
    other = {}
    for r in rows[s:e]:
        n = r['Kernel_Name']
        d = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
        tot += d
        if 'conv_tile' in n or 'conv_post' in n or 'conv1d_direct' in n or 'convt1d_direct' in n or 'resblock_pair' in n or 'resblock2_stage' in n or 'conv_bf16' in n or 'stage_bf16' in n\
                or 'conv_split' in n or 'stage_split' in n or 'convt_bf16' in n or 'n16_stage' in n or 'n16s_stage' in n or 'n32s_stage' in n:
            grp = groups[li]; li += 1
            nm = '+'.join(g.replace('resblocks.', 'rb') for g in grp)
            if len(grp) > 3:                   # a whole stage: first .. last
                nm = grp[0].replace('resblocks.', 'rb') + '..' + grp[-1].replace('resblocks.', 'rb')
            l = dict(name=nm, flops=sum(by_name[g]['flops'] for g in grp), bytes=sum(by_name[g]['bytes'] for g in grp))
            short = n.replace('void ', '').replace('(anonymous namespace)::', '').split('(')[0][:52]
            print(f"{l['name']:20s} {short:52s} {d:8.1f} us {l['flops'] / d / 1e6:7.1f} TF {l['bytes'] / d / 1e3:7.0f} GB/s "
                  f"grid={r['Grid_Size_X']} wg={r['Workgroup_Size_X']} lds={r['LDS_Block_Size']} vgpr={r['VGPR_Count']} agpr={r['Accum_VGPR_Count']}")
        else:
            k = n.replace('void ', '').replace('(anonymous namespace)::', '').split('(')[0]
            other[k] = other.get(k, 0) + d
    for k, v in sorted(other.items(), key=lambda kv: -kv[1]):
        print(f'  other {k:40s} {v:8.1f} us')
    print('sum kernel us', round(tot, 1), 'span us', (int(rows[e - 1]['End_Timestamp']) - int(rows[s]['Start_Timestamp'])) / 1e3)


if __name__ == '__main__':
    main(sys.argv[1], *(int(a) for a in sys.argv[2:]))
