#!/usr/bin/env python3
"""HBM traffic per kernel launch from two rocprofv3 PMC passes (FETCH_SIZE and WRITE_SIZE do not
fit one pass on gfx950: 3 + 2 of the 4 TCC slots).

    rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_fetch -o t -- <cmd>
    rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc_write -o t -- <cmd>
    tools/pmc_traffic.py gpurun_out/pmc_fetch gpurun_out/pmc_write > profiles/rNN_traffic.json

Both counters are in KiB.  Correction applied (MI355X_MICROARCH.md, HBM section): on gfx950
FETCH_SIZE tallies the 128-byte requests of wide coalesced streams at 64 bytes, so reads are doubled:
    hbm_bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024
Kernel functions are folded onto the plan kernel names bench.py groups by.
"""
import csv
import glob
import json
import os
import sys

FOLD = [  # (substring of the demangled kernel function, dtype marker, plan kernel name)
    ("conv_igemm_wave_kernel<true", "conv_igemm_wave_i8_mfma32x32x32"),
    ("conv_igemm_wave_kernel<false", "conv_igemm_wave_f16_mfma32x32x16"),
    ("conv_igemm_patch_kernel", "conv_igemm_patch_i8_mfma32x32x32"),
    ("conv_igemm_pc_kernel<true", "conv_igemm_pc_i8_mfma32x32x32"),
    ("conv_igemm_pc_kernel<false", "conv_igemm_pc_f16_mfma32x32x16"),
    ("conv_igemm_pcx_kernel<true", "conv_igemm_pc_i8_mfma32x32x32"),
    ("conv_igemm_pcx_kernel<false", "conv_igemm_pc_f16_mfma32x32x16"),
    ("conv_igemm_res_kernel", "conv_igemm_res_i8_mfma32x32x32"),
    ("conv_gemv_kernel<true", "conv_gemv_i8_dot4"),
    ("conv_gemv_kernel<false", "conv_gemv_f16_fma"),
    ("dwconv_channel_nchw_i8_kernel", "dwconv_channel_nchw_i8"),
    ("conv_igemm_pp_kernel<true", "conv_igemm_pp_i8_mfma32x32x32"),
    ("conv_igemm_pp_kernel<false", "conv_igemm_pp_f16_mfma32x32x16"),
    ("conv_igemm_tile_kernel<true", "conv_igemm_tile_i8_mfma32x32x32"),
    ("conv_igemm_halo_kernel<true", "conv_igemm_tile_i8_mfma32x32x32"),
    ("conv_igemm_halo_kernel<false", "conv_igemm_tile_f16_mfma32x32x16"),
    ("conv_igemm_tile_kernel<false", "conv_igemm_tile_f16_mfma32x32x16"),
    ("conv_igemm_regs_kernel<true", "conv_igemm_regs_i8_mfma32x32x32"),
    ("conv_igemm_regs_kernel<false", "conv_igemm_regs_f16_mfma32x32x16"),
    ("conv_stem_i8_mfma_kernel", "conv_stem_i8_mfma32x32x32"),
    ("conv_stem_i8_kernel", "conv_stem_i8_dot4"),
    ("dwconv_nhwc_kernel<true", "dwconv_nhwc_i8"),
    ("dwconv3x3_i8_dot4_kernel", "dwconv_nhwc_i8"),
    ("dwpw_fused_kernel", "dwpw_fused_i8"),
    ("dwpw_stream_kernel", "dwpw_stream_i8"),
    ("conv1x1_resident_kernel", "conv1x1_resident_i8_mfma32x32x32"),
    ("conv1x1_latency_kernel", "conv1x1_latency_i8_mfma32x32x32"),
    ("dwpw_resident_kernel", "dwpw_resident_i8"),
    ("pwdw_f16_nchw_kernel", "pwdw_f16_nchw"),
    ("stemdw_f16_nchw_kernel", "stemdw_f16_nchw"),
    ("conv_group_direct_kernel", "conv_group_direct"),
    ("pwdw_fused_kernel", "pwdw_fused_i8"),
    ("stemdw_fused_kernel", "stemdw_fused_i8"),
    ("pwdw_stream_kernel", "pwdw_stream_i8"),
    ("conv1x1_stream_kernel", "conv1x1_stream_i8_mfma32x32x32"),
    ("dwconv3x3_i8_mfma_kernel", "dwconv_mfma_i8"),
    ("dwconv_nhwc_kernel<false", "dwconv_nhwc_f16"),
    ("conv_direct_kernel<true", "conv_direct_i8"),
    ("conv_direct_kernel<false", "conv_direct_f16"),
    ("transpose_", "layout_transpose"),
]


def fold(name):
    for sub, plan in FOLD:
        if sub in name:
            return plan
    return name.split("(")[0]


def read(directory, counter):
    """-> {plan kernel name: [sum KiB, dispatches]}"""
    acc = {}
    files = glob.glob(os.path.join(directory, "**", "*counter_collection.csv"), recursive=True)
    if not files:
        raise SystemExit("no *counter_collection.csv under %s" % directory)
    for path in files:
        with open(path, newline="") as f:
            for row in csv.DictReader(f):
                if not row["Counter_Name"].startswith(counter):
                    continue
                a = acc.setdefault(fold(row["Kernel_Name"]), [0.0, 0])
                a[0] += float(row["Counter_Value"])
                a[1] += 1
    return acc


def main(fetch_dir, write_dir):
    fetch, write = read(fetch_dir, "FETCH_SIZE"), read(write_dir, "WRITE_SIZE")
    out = {"_doc": "HBM bytes per launch = (2*FETCH_SIZE + WRITE_SIZE)*1024, averaged over the dispatches of "
                   "each kernel function in two separate rocprofv3 --pmc passes of the same command"}
    for k in sorted(set(fetch) | set(write)):
        fk, fn = fetch.get(k, [0.0, 0])
        wk, wn = write.get(k, [0.0, 0])
        f_avg = fk / fn if fn else 0.0
        w_avg = wk / wn if wn else 0.0
        out[k] = {"dispatches": max(fn, wn), "fetch_kib_raw": f_avg, "write_kib_raw": w_avg,
                  "hbm_bytes_per_launch": (2.0 * f_avg + w_avg) * 1024.0}
    json.dump(out, sys.stdout, indent=1, sort_keys=True)
    sys.stdout.write("\n")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
