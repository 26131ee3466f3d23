#!/usr/bin/env python
"""Fold one rocprofv3 --pmc pass (SQ / GRBM counters, csv) into per-kernel per-launch averages with the derived figures the
roofline discussion uses (MI355X_MICROARCH.md: SQ_VALU_MFMA_BUSY_CYCLES counts cycles, SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_*
quad-cycles, GRBM_GUI_ACTIVE is summed over the 8 XCDs):
    clock_GHz      = GRBM_GUI_ACTIVE / 8 / duration
    mfma_busy_frac = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE / 8)
    python tools/pmc_sq_summary.py gpurun_out/pmc_dir profiles/<name>.json [kernel-name-substring ...]
"""
import csv
import glob
import json
import os
import re
import sys


def main():
    root, dst, filt = sys.argv[1], sys.argv[2], sys.argv[3:]
    per = {}
    for path in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
        with open(path) as f:
            for row in csv.DictReader(f):
                name = re.sub(r"\(.*$", "", row["Kernel_Name"]).strip()
                if filt and not any(s in name for s in filt):
                    continue
                key = (name, row["Grid_Size"], row["Dispatch_Id"])
                d = per.setdefault(key, {"dur_us": (int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1e3,
                                         "vgpr": int(row["VGPR_Count"]) + int(row["Accum_VGPR_Count"]), "lds": int(row["LDS_Block_Size"])})
                d[row["Counter_Name"]] = d.get(row["Counter_Name"], 0.0) + float(row["Counter_Value"])
    groups = {}
    for (name, grid, _), d in per.items():
        groups.setdefault(f"{name} grid={grid}", []).append(d)
    out = {}
    for k, lst in sorted(groups.items()):
        avg = {c: sum(d.get(c, 0.0) for d in lst) / len(lst) for c in lst[0]}
        avg["launches"] = len(lst)
        g = avg.get("GRBM_GUI_ACTIVE")
        if g:
            avg["clock_GHz"] = g / 8.0 / (avg["dur_us"] * 1e3)
            if "SQ_VALU_MFMA_BUSY_CYCLES" in avg:
                avg["mfma_busy_frac"] = avg["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024.0 * g / 8.0)
        w = avg.get("SQ_WAVE_CYCLES")
        if w:
            for c in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_LDS", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_MISC"):
                if c in avg:
                    avg[c + "_per_wave_cycle"] = avg[c] / w
        out[k] = {a: (round(b, 4) if isinstance(b, float) else b) for a, b in avg.items()}
    with open(dst, "w") as f:
        json.dump(out, f, indent=1)
    for k, v in out.items():
        print(k[:110], "dur_us", v["dur_us"], "clock", v.get("clock_GHz"), "mfma_busy", v.get("mfma_busy_frac"))


if __name__ == "__main__":
    main()
