#!/usr/bin/env python
"""Fold two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE: they do not fit one pass on gfx950, see
MI355X_MICROARCH.md) into per-kernel per-launch averages -> profiles/<name>_pmc_hbm_traffic.json.

    rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_fetch -- python bench.py --steps 1 --warmup 0 --ddim-steps 2 --no-cpu-baseline --no-roofline
    rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc_write -- python bench.py --steps 1 --warmup 0 --ddim-steps 2 --no-cpu-baseline --no-roofline
    python tools/pmc_summary.py gpurun_out/pmc_fetch gpurun_out/pmc_write profiles/round1_e_pmc_hbm_traffic.json

Values are the counters as reported (KB); bench.py applies the guide's gfx950 correction (FETCH_SIZE x 2
for wide coalesced reads) when it folds the conv kernels into roofline.traffic.
"""
import csv
import glob
import json
import os
import re
import sys


def fold(root, counter):
    out = {}
    for path in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
        with open(path) as f:
            for row in csv.DictReader(f):
                if row.get("Counter_Name") != counter:
                    continue
                name = re.sub(r"\(.*$", "", row["Kernel_Name"]).strip()
                e = out.setdefault(name, [0, 0.0])
                e[0] += 1
                e[1] += float(row["Counter_Value"])
    return out


def main():
    fetch_dir, write_dir, dst = sys.argv[1:4]
    fe, wr = fold(fetch_dir, "FETCH_SIZE"), fold(write_dir, "WRITE_SIZE")
    res = {}
    for name in sorted(set(fe) | set(wr)):
        n = fe.get(name, wr.get(name))[0]
        res[name] = {"launches": n,
                     "FETCH_SIZE_KB_avg": fe[name][1] / fe[name][0] if name in fe else None,
                     "WRITE_SIZE_KB_avg": wr[name][1] / wr[name][0] if name in wr else None}
    with open(dst, "w") as f:
        json.dump(res, f, indent=1)
    print(f"wrote {dst}: {len(res)} kernels")


if __name__ == "__main__":
    main()
