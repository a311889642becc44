#!/usr/bin/env python3
"""Per-kernel SQ counters from rocprofv3 --pmc passes over bench.py -> profiles/rNN_sq_counters.json.

    rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES \
              SQ_INSTS_MFMA GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/sq_a -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-alt
    rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE \
              SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU --output-format csv -d gpurun_out/sq_b -- (same)
    python tools/pmc_sq.py gpurun_out/sq_a gpurun_out/sq_b > profiles/r02_sq_counters.json

Derived per kernel (MI355X_MICROARCH.md: SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles, SQ_VALU_MFMA_BUSY_CYCLES counts
cycles, GRBM_GUI_ACTIVE is summed over the 8 XCDs):
  mfma_busy          = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 * 1024 SIMDs)        share of SIMD-cycles the matrix pipe is busy
  wait_inst_any      = SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES        issue stalls (mostly: the pipe is taken)
  wait_any           = SQ_WAIT_ANY / SQ_WAVE_CYCLES             parked in s_waitcnt / barriers
  lds_bank_conflict  = SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE
  valu_per_mfma      = (SQ_INSTS_VALU - SQ_INSTS_MFMA) / SQ_INSTS_MFMA      vector instructions issued per MFMA (they share the ALU)
"""
import csv
import glob
import json
import os
import sys


def per_kernel(path):
    fs = glob.glob(os.path.join(path, '**', '*_counter_collection.csv'), recursive=True)
    acc = {}
    for f in fs:
        for r in csv.DictReader(open(f)):
            name = r['Kernel_Name'].replace('void ', '').replace('(anonymous namespace)::', '').split('(')[0]
            d = acc.setdefault(name, {})
            d[r['Counter_Name']] = d.get(r['Counter_Name'], 0.0) + float(r['Counter_Value'])
            if r['Counter_Name'] == 'SQ_WAVE_CYCLES' or r['Counter_Name'] == 'SQ_INSTS_VALU':
                d['_launches_' + r['Counter_Name']] = d.get('_launches_' + r['Counter_Name'], 0) + 1
    return acc


def main(dirs):
    merged = {}
    for d in dirs:
        for k, v in per_kernel(d).items():
            merged.setdefault(k, {}).update(v)
    out = {'_meta': dict(commit=os.environ.get('V2W_COMMIT', 'unknown'), date=os.environ.get('V2W_DATE', 'unknown'),
                         command='bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-alt under rocprofv3 --pmc (two passes)')}
    for k, c in sorted(merged.items(), key=lambda kv: -kv[1].get('SQ_VALU_MFMA_BUSY_CYCLES', 0)):
        g = lambda n: c.get(n, 0.0)
        if g('SQ_WAVE_CYCLES') <= 0:
            continue
        d = {n: v for n, v in c.items() if not n.startswith('_')}
        simd_cycles = g('GRBM_GUI_ACTIVE') / 8 * 1024
        d['derived'] = dict(
            mfma_busy=g('SQ_VALU_MFMA_BUSY_CYCLES') / simd_cycles if simd_cycles else None,
            wait_inst_any=g('SQ_WAIT_INST_ANY') / g('SQ_WAVE_CYCLES'),
            wait_any=g('SQ_WAIT_ANY') / g('SQ_WAVE_CYCLES'),
            active_inst_any=g('SQ_ACTIVE_INST_ANY') / g('SQ_WAVE_CYCLES'),
            lds_bank_conflict=(g('SQ_LDS_BANK_CONFLICT') / g('SQ_LDS_IDX_ACTIVE')) if g('SQ_LDS_IDX_ACTIVE') else None,
            valu_per_mfma=((g('SQ_INSTS_VALU') - g('SQ_INSTS_MFMA') * (c.get('_launches_SQ_INSTS_VALU', 1) / max(1, c.get('_launches_SQ_WAVE_CYCLES', 1))))
                           / (g('SQ_INSTS_MFMA') * (c.get('_launches_SQ_INSTS_VALU', 1) / max(1, c.get('_launches_SQ_WAVE_CYCLES', 1)))))
            if g('SQ_INSTS_MFMA') and g('SQ_INSTS_VALU') else None,
            launches=c.get('_launches_SQ_WAVE_CYCLES'))
        out[k] = d
    json.dump(out, sys.stdout, indent=1)


if __name__ == '__main__':
    main(sys.argv[1:])
