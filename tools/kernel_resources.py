#!/usr/bin/env python3
"""Registers / LDS / scratch of every kernel of one HIP source, from hipcc's gfx950 assembly (no GPU needed).

    python tools/kernel_resources.py wavthruvec_pytorch_amd/csrc/v2w_conv_mfma.hip [name-filter]

Prints per kernel: arch VGPRs, AGPRs, total (the unified-file allocation that sets waves/SIMD), SGPRs, static LDS, scratch.
"""
import re
import subprocess
import sys
import tempfile


def main():
    src = sys.argv[1]
    filt = sys.argv[2] if len(sys.argv) > 2 else ''
    with tempfile.NamedTemporaryFile(suffix='.s') as f:
        subprocess.run(['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-S', '--cuda-device-only',
                        '-o', f.name, src], check=True, stderr=subprocess.DEVNULL)
        text = open(f.name).read()
    if len(sys.argv) > 3:
        open(sys.argv[3], 'w').write(text)
    for m in re.finditer(r'- \.agpr_count:\s+(\d+).*?\.group_segment_fixed_size:\s+(\d+).*?\.name:\s+(\S+).*?'
                         r'\.private_segment_fixed_size:\s+(\d+).*?\.sgpr_count:\s+(\d+).*?\.vgpr_count:\s+(\d+)', text, re.S):
        agpr, lds, name, scratch, sgpr, vgpr = m.groups()
        dem = subprocess.run(['c++filt', name], capture_output=True, text=True).stdout.strip()
        dem = dem.replace('(anonymous namespace)::', '').replace('void ', '').split('(')[0]
        if filt in dem:
            tot = int(vgpr)
            alloc = (tot + 7) // 8 * 8
            waves = min(8, 512 // alloc) if alloc else 8
            print(f'{dem:70s} vgpr+agpr={tot:3d} (agpr {agpr:>3s}) waves/SIMD={waves} sgpr={sgpr:>3s} lds={lds:>6s} scratch={scratch}')


if __name__ == '__main__':
    main()
