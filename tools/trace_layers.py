#!/usr/bin/env python3
"""Per-layer view of one benchmark iteration from a rocprofv3 --kernel-trace CSV (1024^2 VGG19 workload)."""
import csv
import sys

path = sys.argv[1]
size = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
rows = list(csv.DictReader(open(path)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if 'image_pass' in r['Kernel_Name']]
which = int(sys.argv[3]) if len(sys.argv) > 3 else len(idx) // 2
it = rows[idx[which - 1] + 1: idx[which] + 1]
fw = [('conv1_1', 3, 64, 1), ('conv1_2', 64, 64, 1), ('conv2_1', 64, 128, 2), ('conv2_2', 128, 128, 2),
      ('conv3_1', 128, 256, 4), ('conv3_2', 256, 256, 4), ('conv3_3', 256, 256, 4), ('conv3_4', 256, 256, 4),
      ('conv4_1', 256, 512, 8), ('conv4_2', 512, 512, 8), ('conv4_3', 512, 512, 8), ('conv4_4', 512, 512, 8),
      ('conv5_1', 512, 512, 16)]
convs = [r for r in it if 'conv3x3' in r['Kernel_Name']]


def wgs(r):
    """workgroups of a launch: the grid may be two- or three-dimensional (rocprofv3 reports work-items per dimension)"""
    n = 1
    for ax in 'XYZ':
        g, w = int(r.get('Grid_Size_' + ax, 1) or 1), int(r.get('Workgroup_Size_' + ax, 1) or 1)
        n *= max(1, g // max(1, w))
    return n


names = [f[0] + ' fwd' for f in fw] + [f[0] + ' bwd' for f in reversed(fw)]
specs = fw + list(reversed(fw))
tot = totf = 0
for r, n, s in zip(convs, names, specs):
    d = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
    px = (size // s[3]) ** 2
    fl = 2 * 9 * s[1] * s[2] * px
    tot += d
    totf += fl
    kn = r['Kernel_Name'].split('(')[0].replace('void st2::', '')[:34]
    print('%-12s %-34s wgs=%-6d %8.1f us %6.1f TF/s  vgpr=%s+%s lds=%s' % (
        n, kn, wgs(r), d, fl / d / 1e6, r['VGPR_Count'],
        r['Accum_VGPR_Count'], r['LDS_Block_Size']))
print('conv total %.1f us, %.1f TF/s' % (tot, totf / tot / 1e6))
print('--- every kernel of the iteration (duration, gap to previous)')
prev = int(rows[idx[which - 1]]['End_Timestamp'])      # end of the previous iteration's last kernel
other = busy = gaps = 0.0
for r in it:
    s_, e_ = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    gap = (s_ - prev) / 1e3                              # idle time on the stream in front of this kernel (negative: overlap)
    prev = max(prev, e_)
    busy += (e_ - s_) / 1e3
    gaps += max(gap, 0.0)
    kn = r['Kernel_Name']
    if 'conv3x3' not in kn or 'dgrad_smallM' in kn or 'dgrad_first' in kn:
        other += (e_ - s_) / 1e3
        print('%-44s %9.1f us gap %6.2f' % (kn.split('(')[0].replace('void st2::', '')[:44], (e_ - s_) / 1e3, gap))
span = (int(it[-1]['End_Timestamp']) - int(rows[idx[which - 1]]['End_Timestamp'])) / 1e3
if gaps == 0.0:
    # rocprofv3's Start_Timestamp of a kernel queued behind another on the same stream is the moment the previous one ended (the
    # dispatch is already waiting on it), so the ~1.5 us between the last wave of one kernel and the first wave of the next sits
    # INSIDE the next kernel's duration.  What this does verify: sum of durations == span, i.e. the host never let the stream run dry
    # (a host-side stall would show as a positive gap).  Compare span with bench.py's ms_per_step to close the loop.
    print('(gap 0.00 everywhere: back-to-back dispatches -- the inter-kernel bubble is inside each duration; no host-side stall in this iteration)')
print('kernels other than the matrix-core convs %.1f us; all kernels %.1f us + gaps %.1f us = iteration span %.1f us (%d launches, %.2f us per gap)' % (
    other, busy, gaps, span, len(it), gaps / max(1, len(it))))
