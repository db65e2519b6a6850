#!/usr/bin/env python3
"""Copy one measurement set out of gpurun_out/ into profiles/ and patch the headline numbers in the docs.

    python tools/refresh_profiles.py <bench.log> <rocprof stats dir> <pmc fetch dir> <pmc write dir>
"""
import csv
import glob
import json
import re
import subprocess
import sys

ROOT = __file__.rsplit("/tools/", 1)[0]
bench_log, prof_dir, fetch_dir, write_dir = sys.argv[1:5]
subprocess.check_call([sys.executable, ROOT + "/tools/pmc_summary.py", fetch_dir, write_dir, ROOT + "/profiles/r01_pmc_c2.json"],
                      stdout=subprocess.DEVNULL)
line = [x for x in open(bench_log) if x.startswith("{")][-1]
open(ROOT + "/profiles/r01_bench_c2.json", "w").write(line)
b = json.loads(line)
stats = glob.glob(prof_dir + "/**/*kernel_stats.csv", recursive=True)[0]
open(ROOT + "/profiles/r01_c2_train_kernel_stats.csv", "w").write(open(stats).read())
rows = list(csv.DictReader(open(stats)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
pm = json.load(open(ROOT + "/profiles/r01_pmc_c2.json"))
agg = [v for k, v in pm["kernels"].items() if "segsum" in k and "bwd" not in k][0]
with open(ROOT + "/profiles/r01_c2_train_kernel_stats.md", "w") as f:
    f.write("# Round 1, final state: `rocprofv3 --kernel-trace --stats -- python3 bench.py --no-cpu` (c2; default mode = "
            "training step, the run also times the forward-only pass and, as a side field, both with hoist_message)\n\n26 forward-only passes + "
            "26 training passes (3 warm-up + 10 timed each; half of them with BasicModel.hoist_message, i.e. message + aggregate once "
            "per pass instead of per step), 3 message-passing steps per pass, plus 13 launches of the stream calibration.\n\n| kernel | calls | avg ms | % of GPU time |\n|---|---|---|---|\n")
    for r in rows[:12]:
        f.write("| `%s` | %s | %.3f | %.1f |\n" % (r["Name"].split("(")[0][:80], r["Calls"], float(r["AverageNs"]) / 1e6,
                                                  100 * float(r["TotalDurationNs"]) / tot))
    f.write("\nUn-profiled `python bench.py` on the same commit (`profiles/r01_bench_c2.json`): training step %.2f ms = %.3f G "
            "edges/s, forward pass %.2f ms = %.3f G edges/s; aggregator live HIP-event average %.3f ms -> %.0f GB/s "
            "algorithmic = %.1f%% of 8 TB/s; CPU baseline %.0f edges/s on %d threads.\n"
            % (b["ms_per_step"], b["value"] / 1e9, b["forward"]["ms_per_step"], b["forward"]["value"] / 1e9,
               b["roofline"]["avg_launch_ms"], b["roofline"]["achieved"], 100 * b["roofline"]["frac"],
               b["cpu_baseline"]["value"], b["cpu_baseline"]["cores"]))
    f.write("\nPMC traffic per launch (`profiles/r01_pmc_c2.json`: FETCH_SIZE and WRITE_SIZE in separate `--pmc` passes, "
            "KiB -> bytes, FETCH x2 for gfx950's half-counted wide reads):\n\n| kernel | HBM read GB | HBM write GB |\n|---|---|---|\n")
    for k, v in pm["kernels"].items():
        f.write("| `%s` | %.3f | %.3f |\n" % (k[:70], v["hbm_read_bytes"] / 1e9, v["hbm_write_bytes"] / 1e9))
    f.write("\nAggregator: algorithmic bytes %.3f GB per launch, PMC traffic %.3f GB -> traffic/algorithmic = %.3f.\n"
            % (b["roofline"]["algorithmic_bytes_per_launch"] / 1e9, agg["hbm_bytes"] / 1e9,
               agg["hbm_bytes"] / b["roofline"]["algorithmic_bytes_per_launch"]))
    cal = b["roofline"].get("stream_calibration")
    if cal:
        f.write("Calibration in the same bench run: %s reaches %.0f GB/s (%.3f ms).\n" % (cal["op"], cal["GB/s"], cal["ms"]))
    try:
        sq = json.load(open(ROOT + "/profiles/r01_sq_c2.json"))["kernels"]
        f.write("\nSQ counters (`profiles/r01_sq_c2.json`, one `--pmc` pass; fractions of wave cycles, MFMA pipe busy as a "
                "fraction of the kernel's duration per SIMD):\n\n| kernel | parked on s_waitcnt/barrier | issue stall | issuing | MFMA busy |\n|---|---|---|---|---|\n")
        for k, v in sq.items():
            f.write("| `%s` | %.2f | %.2f | %.2f | %s |\n" % (k[:70], v["wave_parked_frac"], v["issue_stall_frac"], v["issuing_frac"],
                                                             "%.2f" % v["mfma_busy_frac"] if v["mfma_busy_frac"] is not None else "n/a"))
    except FileNotFoundError:
        pass
    f.write("\nOther workloads (`profiles/r01_bench_c{3,4,5}.json`, same commit; all widths on the bf16x6 kernels: resident slices at 128, "
            "streamed weights at 256):\n\n| workload | training step ms | forward pass ms | aggregator frac of 8 TB/s |\n|---|---|---|---|\n")
    for w in ("c3", "c4", "c5"):
        d = json.load(open(ROOT + "/profiles/r01_bench_%s.json" % w))
        f.write("| %s | %.1f | %.1f | %.3f |\n" % (d["config"]["workload"][:70], d["ms_per_step"], d["forward"]["ms_per_step"], d["roofline"]["frac"]))
for path in ("DESIGN.md", "README.md"):
    s = open(ROOT + "/" + path).read()
    s = re.sub(r"training step \(forward \+ backward \+ flat-gradient all-reduce\) [0-9.]+ ms =\n\*\*[0-9.]+ G edges/s\*\*; forward-only pass [0-9.]+ ms = \*\*[0-9.]+ G edges/s\*\*",
               "training step (forward + backward + flat-gradient all-reduce) %.1f ms =\n**%.2f G edges/s**; forward-only pass %.2f ms = **%.2f G edges/s**"
               % (b["ms_per_step"], b["value"] / 1e9, b["forward"]["ms_per_step"], b["forward"]["value"] / 1e9), s)
    s = re.sub(r"Round-1 numbers \(1× MI355X, c2\): training [0-9.]+ G edges/s \([0-9.]+ ms / pass\); forward [0-9.]+ G edges/s \([0-9.]+ ms / pass\)\.",
               "Round-1 numbers (1× MI355X, c2): training %.2f G edges/s (%.1f ms / pass); forward %.2f G edges/s (%.2f ms / pass)."
               % (b["value"] / 1e9, b["ms_per_step"], b["forward"]["value"] / 1e9, b["forward"]["ms_per_step"]), s)
    s = re.sub(r"Round-1 numbers on one MI355X \(c2\): training step [0-9.]+ G edges/s, forward [0-9.]+ G edges/s, aggregator [0-9.]+ TB/s\n\([0-9]+ % of the 8 TB/s HBM peak, PMC traffic [0-9.]+× algorithmic\)",
               "Round-1 numbers on one MI355X (c2): training step %.2f G edges/s, forward %.2f G edges/s, aggregator %.1f TB/s\n(%d %% of the 8 TB/s HBM peak, PMC traffic %.3f× algorithmic)"
               % (b["value"] / 1e9, b["forward"]["value"] / 1e9, b["roofline"]["achieved"] / 1e3, round(100 * b["roofline"]["frac"]),
                  agg["hbm_bytes"] / b["roofline"]["algorithmic_bytes_per_launch"]), s)
    open(ROOT + "/" + path, "w").write(s)
print("train %.2f ms (%.3f G/s)  fwd %.2f ms (%.3f G/s)  agg %.0f GB/s frac %.3f  traffic ratio %.3f" % (
    b["ms_per_step"], b["value"] / 1e9, b["forward"]["ms_per_step"], b["forward"]["value"] / 1e9, b["roofline"]["achieved"],
    b["roofline"]["frac"], agg["hbm_bytes"] / b["roofline"]["algorithmic_bytes_per_launch"]))
