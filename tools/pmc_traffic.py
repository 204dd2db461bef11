#!/usr/bin/env python3
"""HBM traffic per kernel class from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) of bench.py.

Units and corrections per /opt/skills/guides/MI355X_MICROARCH.md (HBM / rocprofv3): the counters are in KiB;
on gfx950 FETCH_SIZE reports exactly half of the bytes of wide (16 B/lane) coalesced streaming reads, so
the read side is doubled for kernels whose loads are dwordx4 / LDS-DMA dwordx4; WRITE_SIZE is exact.
Usage: pmc_traffic.py <fetch counter csv> <write counter csv> [out.json]"""
import collections
import csv
import json
import sys


def per_kernel(path, counter):
    acc = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(path)):
        if r['Counter_Name'] != counter:
            continue
        name = r['Kernel_Name'].split('(')[0].replace('void ', '').replace('st2::', '')
        acc[name][0] += 1
        acc[name][1] += float(r['Counter_Value'])
    return acc


fetch = per_kernel(sys.argv[1], 'FETCH_SIZE')
write = per_kernel(sys.argv[2], 'WRITE_SIZE')
out = {}
conv_n = conv_rd = conv_wr = 0
for name in sorted(set(fetch) | set(write)):
    n = fetch[name][0] or write[name][0]
    rd = fetch[name][1] * 1024 / max(1, fetch[name][0])
    wr = write[name][1] * 1024 / max(1, write[name][0])
    wide = ('mfma' in name) or ('wino' in name) or ('rocclr' in name)          # dwordx4 / LDS-DMA x4 readers
    rd_corr = rd * (2 if wide else 1)
    out[name] = dict(launches=n, fetch_bytes_per_launch_raw=rd, fetch_bytes_per_launch=rd_corr,
                     write_bytes_per_launch=wr, read_correction='x2 (gfx950 wide loads)' if wide else 'none')
    if name.startswith('conv3x3_mfma') or name.startswith('conv3x3_wino'):
        conv_n += n; conv_rd += rd_corr * n; conv_wr += wr * n
    print('%-44s n=%-4d read %8.1f MB (raw %8.1f)  write %8.1f MB' % (name[:44], n, rd_corr / 1e6, rd / 1e6, wr / 1e6))
out['_conv3x3_all'] = dict(launches=conv_n, hbm_bytes_per_launch=(conv_rd + conv_wr) / max(1, conv_n),
                                    read_bytes_per_launch=conv_rd / max(1, conv_n), write_bytes_per_launch=conv_wr / max(1, conv_n))
print('conv3x3 matrix-core launches (direct + Winograd): %.1f MB read + %.1f MB written per launch (avg over %d launches)' % (
    conv_rd / max(1, conv_n) / 1e6, conv_wr / max(1, conv_n) / 1e6, conv_n))
if len(sys.argv) > 3:
    import os
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    out['_meta'] = dict(source_sha16=bench.source_hash(), profile=os.environ.get('ST2_PROFILE_TAG', '?'),
                        note='HBM bytes per launch from separate --pmc FETCH_SIZE / WRITE_SIZE passes; FETCH_SIZE doubled for wide loads')
    json.dump(out, open(sys.argv[3], 'w'), indent=1)
