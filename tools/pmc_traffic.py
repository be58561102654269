#!/usr/bin/env python
"""Aggregates the two rocprofv3 PMC passes of tools/profile_hbm_traffic.sh into
    profiles/<tag>_pmc_hbm_traffic.csv     per kernel: launches, FETCH_SIZE / WRITE_SIZE averages, corrected bytes
    profiles/pmc_traffic.json              HBM bytes per launch of the kernel families bench.py reports, stamped with
                                           the hash of the kernel sources (bench.py emits `traffic: null` on a mismatch)
Corrections as MI355X_MICROARCH.md (HBM section) prescribes for gfx950: FETCH_SIZE (KiB) x 2 for 16-B-per-lane streaming
reads; WRITE_SIZE (KiB) as is.     python tools/pmc_traffic.py gpurun_out/pmc_fetch gpurun_out/pmc_write r02_a
"""
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def per_kernel(folder, counter):
    acc = {}
    for path in glob.glob(os.path.join(folder, "**", "*counter_collection.csv"), recursive=True):
        with open(path) as f:
            for r in csv.DictReader(f):
                if r["Counter_Name"] != counter:
                    continue
                a = acc.setdefault(r["Kernel_Name"], [0, 0.0])
                a[0] += 1
                a[1] += float(r["Counter_Value"])
    return {k: (n, s / n) for k, (n, s) in acc.items()}


def main():
    fetch_dir, write_dir, tag = sys.argv[1], sys.argv[2], sys.argv[3]
    fe, wr = per_kernel(fetch_dir, "FETCH_SIZE"), per_kernel(write_dir, "WRITE_SIZE")
    rows = []
    for k in sorted(fe, key=lambda k: -fe[k][0] * fe[k][1]):
        n, f_kib = fe[k]
        w_kib = wr.get(k, (0, 0.0))[1]
        rows.append((k, n, round(f_kib, 1), int(f_kib * 1024 * 2), round(w_kib, 1), int(w_kib * 1024)))
    out_csv = os.path.join(ROOT, "profiles", "%s_pmc_hbm_traffic.csv" % tag)
    with open(out_csv, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["kernel", "launches", "FETCH_SIZE_avg_KiB", "fetch_bytes_corrected_x2", "WRITE_SIZE_avg_KiB", "write_bytes"])
        w.writerows(rows)
    import bench
    fam = {}
    for k, n, _, fb, _, wb in rows:
        if "gemm_f32_kernel<128, 128, 32, false, false, 0, true" in k:      # (+ the trailing ping-pong flag)
            fam["gemm_wgrad"] = fb + wb
        for key, pat in (("attn_bwd", "attn16_bwd_kernel"), ("attn_fwd", "attn16_fwd_kernel"), ("embed_fwd", "embed_fwd_kernel"),
                         ("embed_bwd", "embed_bwd_kernel"), ("loss", "layout_loss_kernel"), ("adam", "adam_kernel")):
            if pat in k and key not in fam:
                fam[key] = fb + wb
    pair = [(n, fb + wb) for k, n, _, fb, _, wb in rows if "gemm_pair_kernel<" in k]
    if pair:        # a projection's data gradient + weight gradient in one launch: the family's launch-weighted average (two instantiations)
        fam["gemm_pair"] = int(sum(n * b for n, b in pair) / sum(n for n, _ in pair))
    rec = {"_note": "HBM bytes per launch from rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE (separate passes) over bench.py "
                    "--steps 2 --warmup 1; FETCH_SIZE doubled as MI355X_MICROARCH.md (HBM section) prescribes for 16-B-per-lane "
                    "streaming reads on gfx950, WRITE_SIZE as read. Source: profiles/%s_pmc_hbm_traffic.csv" % tag,
           "source_sha16": bench.kernel_source_hash(), "sources": list(bench.TRAFFIC_SOURCES)}
    rec.update(fam)
    with open(os.path.join(ROOT, "profiles", "pmc_traffic.json"), "w") as f:
        json.dump(rec, f, indent=1)
    print(json.dumps(rec, indent=1))


if __name__ == "__main__":
    main()
