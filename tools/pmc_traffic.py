#!/usr/bin/env python3
"""HBM traffic per kernel class from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) of bench.py.

Units and corrections per /opt/skills/guides/MI355X_MICROARCH.md (HBM / rocprofv3): the counters are in KiB;
on gfx950 FETCH_SIZE reports exactly half of the bytes of wide (16 B/lane) coalesced streaming reads, so
the read side is doubled for kernels whose dominant loads are 16 bytes per lane; WRITE_SIZE is exact.  Which kernels
those are is stated per kernel in LOAD_WIDTH below (read off the sources, not guessed from the name); other widths are
uncalibrated by the guide and are reported raw, flagged as such.
Usage: pmc_traffic.py <fetch counter csv> <write counter csv> [out.json]"""
import collections
import csv
import json
import sys


# kernel-name prefix -> bytes per lane of the loads that carry its HBM read traffic (first match wins)
LOAD_WIDTH = (
    ('conv3x3_wino_f32_128x128_anyw', 'mixed'),   # activations: dword LDS-DMA pieces; U stream: dwordx4
    ('conv3x3_wino_f32_64x256_anyw', 'mixed'),
    ('conv3x3_wino', 16),                         # raw_ptr_buffer_load_lds x4 (activations) + global_load_dwordx4 (U), float4 epilogue
    ('conv3x3_first_split_k', 4),                 # the fp32 image, dword loads into the halo tile
    ('conv3x3_mfma', 16),                         # LDS-DMA dwordx4 for weights and activation tiles (fp32 and bf16 kernels)
    ('conv3x3_dgrad_first_f32q', 16), ('conv3x3_dgrad_first_bf16', 16), ('conv3x3_dgrad_first_f32', 4),
    ('conv3x3_dgrad_smallM16', 16), ('conv3x3_dgrad_smallM_dma', 16), ('conv3x3_dgrad_smallM', 4),
    ('wino_combine_k', 16),
    ('gram_partial_dma', 16), ('gram16_partial', 16), ('gram_partial_k', 4), ('gram_fold_k', 4), ('gram_reduce', 4),
    ('style_grad_mfma', 16), ('style_grad_big_k', 16), ('style_grad16', 16), ('style16_pack', 4), ('style_s2_trace_k', 16),
    ('layer_elem_k', 16),
    ('image_pass_k', 16), ('image_pass_tile_k', 4),
    ('maxpool_fwd_v4_k', 16), ('maxpool_bwd_v4_k', 16), ('maxpool_bwd_amap_k', 8), ('maxpool_bwd_idx16_k', 16),
    ('maxpool_fwd_k', 4), ('maxpool_bwd_k', 4),
    ('pack_act16_k', 4),
    ('lbfgs_', 16),
    ('scaled_accumulate_k', 4), ('vec_', 4), ('lincomb_k', 4), ('strip_copy_k', 4),
    ('preprocess_k', 4), ('deprocess_k', 4), ('resample_', 4), ('clamp0_copy_k', 4),
    ('__amd_rocclr_copyBuffer', 16), ('__amd_rocclr_fillBuffer', 16),
)


def load_width(name):
    for prefix, width in LOAD_WIDTH:
        if name.startswith(prefix):
            return width
    return None


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
    width = load_width(name)
    wide = width == 16
    rd_corr = rd * (2 if wide else 1)
    how = ('x2 (16-byte loads: gfx950 FETCH_SIZE counts half)' if wide else
           'none (%s-byte loads: uncalibrated, raw counter)' % width if width else 'none (kernel not in LOAD_WIDTH: raw counter)')
    out[name] = dict(launches=n, fetch_bytes_per_launch_raw=rd, fetch_bytes_per_launch=rd_corr,
                     write_bytes_per_launch=wr, load_bytes_per_lane=width, read_correction=how)
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
