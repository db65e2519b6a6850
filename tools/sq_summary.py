#!/usr/bin/env python3
"""Per-kernel issue/stall/MFMA utilisation from one rocprofv3 SQ counter pass of bench.py.

    rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY \
        SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_VALU_MFMA_MOPS_BF16 \
        -d gpurun_out/pmc_sq --output-format csv -- python3 bench.py --steps 2 --warmup 1 --no-cpu
    python tools/sq_summary.py gpurun_out/pmc_sq profiles/r02_sq_c2.json

Units (MI355X_MICROARCH.md "rocprofv3 PMC slots"): SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* are quad-cycles
summed over waves; WAIT_ANY (parked on s_waitcnt/barrier) + WAIT_INST_ANY (issue stall) + ACTIVE_INST_ANY ~= WAVE_CYCLES.
SQ_VALU_MFMA_BUSY_CYCLES is in cycles summed over the chip's 1024 SIMDs (checked: the fp32 message kernel issues
187.5k tiles x 64 MFMAs x 64 cycles = 7.68e8, the counter reads 7.67e8); GRBM_GUI_ACTIVE is summed over the 8 XCDs
(1.34e7 for a 0.78 ms kernel = 8 x 1.67e6 cycles at ~2.15 GHz).  So
    mfma_busy_frac = MFMA_BUSY / (1024 SIMDs x GUI_ACTIVE / 8)
= the fraction of the kernel's duration the average SIMD's matrix pipe was busy."""
import csv
import glob
import json
import sys


def main():
    f = glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)[0]
    acc = {}
    for r in csv.DictReader(open(f)):
        if "mpnn::" not in r["Kernel_Name"]:
            continue
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        d = acc.setdefault(k, {})
        d.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
    out = {}
    for k, d in acc.items():
        m = {c: sum(v) / len(v) for c, v in d.items()}
        wc = m.get("SQ_WAVE_CYCLES", 0.0) or 1.0
        row = {"launches": len(next(iter(d.values()))), "raw": m,
               "wave_parked_frac": m.get("SQ_WAIT_ANY", 0.0) / wc,
               "issue_stall_frac": m.get("SQ_WAIT_INST_ANY", 0.0) / wc,
               "issuing_frac": m.get("SQ_ACTIVE_INST_ANY", 0.0) / wc}
        gui = m.get("GRBM_GUI_ACTIVE")
        row["mfma_busy_frac"] = (m.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (128.0 * gui)) if gui else None
        out[k] = row
        print("%-58s parked %.2f stall %.2f issuing %.2f mfma_busy %s" % (
            k[:58], row["wave_parked_frac"], row["issue_stall_frac"], row["issuing_frac"],
            "%.2f" % row["mfma_busy_frac"] if row["mfma_busy_frac"] is not None else "n/a"))
    json.dump({"source": "rocprofv3 --kernel-trace --pmc <SQ set> GRBM_GUI_ACTIVE -- python3 bench.py --steps 2 --warmup 1 --no-cpu",
               "kernels": out}, open(sys.argv[2], "w"), indent=1)


if __name__ == "__main__":
    main()
