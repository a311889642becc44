#!/usr/bin/env python3
"""Per-kernel HBM traffic from two rocprofv3 --pmc passes over bench.py (FETCH_SIZE and WRITE_SIZE need separate passes).

    rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_fetch -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline
    rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc_write -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline
    python tools/pmc_traffic.py gpurun_out/pmc_fetch gpurun_out/pmc_write > profiles/r01_cfg2_hbm_traffic.json

Units and corrections per MI355X_MICROARCH.md (HBM section): FETCH_SIZE / WRITE_SIZE count KiB; on gfx950 FETCH_SIZE reports
exactly half of the bytes of a wide coalesced streaming read, so it is doubled; WRITE_SIZE is exact for 16-B streaming stores.
"""
import csv
import glob
import json
import os
import sys


def per_kernel(path, counter):
    f = glob.glob(os.path.join(path, '**', '*_counter_collection.csv'), recursive=True)[0]
    acc = {}
    for r in csv.DictReader(open(f)):
        if r['Counter_Name'] != counter:
            continue
        name = r['Kernel_Name'].replace('void ', '').replace('(anonymous namespace)::', '').split('(')[0]
        d = acc.setdefault(name, [0, 0.0])
        d[0] += 1
        d[1] += float(r['Counter_Value'])
    return acc


def main(fetch_dir, write_dir):
    fe, wr = per_kernel(fetch_dir, 'FETCH_SIZE'), per_kernel(write_dir, 'WRITE_SIZE')
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from wavthruvec_pytorch_amd import build
    out = {'_meta': dict(commit=os.environ.get('V2W_COMMIT', 'unknown'), date=os.environ.get('V2W_DATE', 'unknown'),
                         csrc_sha=build.sources_hash(),
                         command='bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-alt under rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE')}
    for k in sorted(set(fe) | set(wr)):
        nf, f = fe.get(k, [0, 0.0])
        nw, w = wr.get(k, [0, 0.0])
        n = max(nf, nw, 1)
        out[k] = dict(launches=n, fetch_bytes_per_launch=2.0 * f * 1024 / max(nf, 1), write_bytes_per_launch=w * 1024 / max(nw, 1),
                      hbm_bytes_per_launch=2.0 * f * 1024 / max(nf, 1) + w * 1024 / max(nw, 1),
                      note='FETCH_SIZE x2 (gfx950 wide-read correction), KiB -> bytes')
    json.dump(out, sys.stdout, indent=1)


if __name__ == '__main__':
    main(sys.argv[1], sys.argv[2])
