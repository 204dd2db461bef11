#!/usr/bin/env python3
"""Matrix-pipe busy fraction per kernel from one rocprofv3 --pmc pass (SQ_VALU_MFMA_BUSY_CYCLES, GRBM_GUI_ACTIVE).

SQ_VALU_MFMA_BUSY_CYCLES counts cycles summed over every SIMD (64 per v_mfma_f32_32x32x2_f32, 32 per
v_mfma_f32_32x32x16_bf16); GRBM_GUI_ACTIVE is the dispatch's active cycles summed over the 8 XCDs
(/opt/skills/guides/MI355X_MICROARCH.md), so   busy = MFMA_BUSY / (GUI_ACTIVE / 8 * 256 CUs * 4 SIMDs).
Usage: pmc_mfma.py <counter csv> [out.json]"""
import collections
import csv
import json
import sys

acc = collections.defaultdict(lambda: collections.defaultdict(float))
calls = collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    name = r['Kernel_Name'].split('(')[0].replace('void ', '').replace('st2::', '')
    acc[name][r['Counter_Name']] += float(r['Counter_Value'])
    if r['Counter_Name'] == 'GRBM_GUI_ACTIVE':
        calls[name] += 1
out = {}
for name in sorted(acc, key=lambda n: -acc[n].get('GRBM_GUI_ACTIVE', 0)):
    gui, busy = acc[name].get('GRBM_GUI_ACTIVE', 0.0), acc[name].get('SQ_VALU_MFMA_BUSY_CYCLES', 0.0)
    if gui <= 0:
        continue
    frac = busy / (gui / 8.0 * 1024.0)
    out[name] = dict(launches=calls[name], mfma_busy_frac=frac, gui_active_cycles_per_launch=gui / 8.0 / max(1, calls[name]))
    print('%-46s n=%-4d MFMA busy %5.1f %%   %9.0f active cycles per launch' % (name[:46], calls[name], 100 * frac, gui / 8.0 / max(1, calls[name])))
if len(sys.argv) > 2:
    import os
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    out['_meta'] = dict(source_sha16=bench.source_hash(), profile=os.environ.get('ST2_PROFILE_TAG', '?'))
    json.dump(out, open(sys.argv[2], 'w'), indent=1)
