#!/usr/bin/env python3
"""Copy one measurement set (tools/measure_all.sh, gpurun_out/m_*) into profiles/r02_*: bench lines, rocprofv3 kernel
statistics, PMC traffic (FETCH_SIZE / WRITE_SIZE passes) and SQ counters, each stamped with the commit it was measured
at.  Together with measure_all.sh this is the only writer of profiles/r02_*.

    python tools/refresh_profiles.py [commit]
"""
import csv
import glob
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = os.path.join(ROOT, "gpurun_out")
P = os.path.join(ROOT, "profiles")
commit = sys.argv[1] if len(sys.argv) > 1 else subprocess.run(["git", "-C", ROOT, "rev-parse", "HEAD"], capture_output=True,
                                                               text=True).stdout.strip()


def bench_line(path):
    lines = [x for x in open(path) if x.startswith("{")]
    return lines[-1] if lines else None


for w in ("c2", "c1", "c3", "c3a", "c4", "c5", "c4strong"):
    f = os.path.join(G, "m_bench_%s.log" % w)
    if os.path.exists(f) and bench_line(f):
        d = json.loads(bench_line(f))
        d["measured_at_commit"] = commit
        open(os.path.join(P, "r02_bench_%s.json" % w), "w").write(json.dumps(d) + "\n")
# PMC traffic and SQ counters
subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "pmc_summary.py"), os.path.join(G, "m_fetch"),
                       os.path.join(G, "m_write"), os.path.join(P, "r02_pmc_c2.json"), commit], stdout=subprocess.DEVNULL)
subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "sq_summary.py"), os.path.join(G, "m_sq"),
                       os.path.join(P, "r02_sq_c2.json")], stdout=subprocess.DEVNULL)
b = json.loads(bench_line(os.path.join(G, "m_bench_c2.log")))
stats = glob.glob(os.path.join(G, "m_prof") + "/**/*kernel_stats.csv", recursive=True)[0]
open(os.path.join(P, "r02_c2_train_kernel_stats.csv"), "w").write(open(stats).read())
rows = list(csv.DictReader(open(stats)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
pm = json.load(open(os.path.join(P, "r02_pmc_c2.json")))
sq = json.load(open(os.path.join(P, "r02_sq_c2.json")))["kernels"]
with open(os.path.join(P, "r02_c2_train_kernel_stats.md"), "w") as f:
    f.write("# Round 2: `rocprofv3 --kernel-trace --stats -- python3 bench.py --no-cpu` (c2, commit %s)\n\n"
            "The default bench run: 3 warm-up + 10 timed forward-only passes, 3 + 10 training passes (forward + backward + the "
            "flat-gradient all-reduce), the same again with BasicModel.hoist_message (a side field), 3 message-passing steps per "
            "pass, plus the stream / standalone-aggregator calibrations.\n\n| kernel | calls | avg ms | %% of GPU time |\n|---|---|---|---|\n"
            % commit[:12])
    for r in rows[:14]:
        f.write("| `%s` | %s | %.3f | %.1f |\n" % (r["Name"].split("(")[0][:90], r["Calls"], float(r["AverageNs"]) / 1e6,
                                                  100 * float(r["TotalDurationNs"]) / tot))
    rf = b["roofline"]
    f.write("\nUn-profiled `python bench.py` on the same commit (`profiles/r02_bench_c2.json`): training step %.2f ms = %.3f G "
            "edges/s, forward pass %.2f ms = %.3f G edges/s.  Aggregator of the timed path = %s: live HIP-event average %.3f ms; "
            "%s = %.3f GB per launch -> %.0f GB/s = %.1f%% of 8 TB/s"
            % (b["ms_per_step"], b["value"] / 1e9, b["forward"]["ms_per_step"], b["forward"]["value"] / 1e9, rf["kernel"].split(" (")[0],
               rf["avg_launch_ms"], rf["formula"], rf["algorithmic_bytes_per_launch"] / 1e9, rf["achieved"], 100 * rf["frac"]))
    if "min_traffic" in rf:
        f.write("; the bytes it must move (%s) = %.3f GB -> %.0f GB/s = %.1f%%"
                % (rf["min_traffic"]["formula"].split(" (")[0], rf["min_traffic"]["bytes"] / 1e9, rf["min_traffic"]["GB/s"],
                   100 * rf["min_traffic"]["frac_of_peak"]))
    f.write(".\n")
    sa = b.get("aggregator_standalone")
    if sa:
        f.write("Standalone segmented-sum aggregator on the same batch (`mpnn_segsum_f32`, %s): %.3f ms -> %.0f GB/s = %.1f%% of 8 TB/s.\n"
                % (sa["formula"], sa["avg_launch_ms"], sa["achieved"], 100 * sa["frac"]))
    cb = b.get("cpu_baseline")
    if cb:
        f.write("CPU baseline (oracle port, %d threads): %.0f edges/s training over %d molecules, %.0f edges/s forward over %d.\n"
                % (cb["cores"], cb["value"], cb["sample_molecules"], cb["forward"]["value"], cb["forward"]["sample_molecules"]))
    f.write("\nPMC traffic per launch (`profiles/r02_pmc_c2.json`: FETCH_SIZE and WRITE_SIZE in separate `--pmc` passes, KiB -> "
            "bytes, FETCH x2 for gfx950's half-counted wide reads):\n\n| kernel | HBM read GB | HBM write GB |\n|---|---|---|\n")
    for k, v in pm["kernels"].items():
        f.write("| `%s` | %.3f | %.3f |\n" % (k[:70], v["hbm_read_bytes"] / 1e9, v["hbm_write_bytes"] / 1e9))
    f.write("\nSQ counters (`profiles/r02_sq_c2.json`, one `--pmc` pass; fractions of wave cycles, MFMA pipe busy as a fraction of "
            "the kernel's duration per SIMD):\n\n| kernel | parked on s_waitcnt/barrier | issue stall | issuing | MFMA busy |\n|---|---|---|---|---|\n")
    for k, v in sq.items():
        f.write("| `%s` | %.2f | %.2f | %.2f | %s |\n" % (k[:70], v["wave_parked_frac"], v["issue_stall_frac"], v["issuing_frac"],
                                                         "%.2f" % v["mfma_busy_frac"] if v["mfma_busy_frac"] is not None else "n/a"))
    f.write("\nOther workloads (`profiles/r02_bench_*.json`, same commit):\n\n| workload | training step ms | forward pass ms | aggregator of the timed path, frac of 8 TB/s |\n|---|---|---|---|\n")
    for w in ("c1", "c3", "c3a", "c4", "c5"):
        pth = os.path.join(P, "r02_bench_%s.json" % w)
        if os.path.exists(pth):
            d = json.loads(open(pth).read())
            if d.get("measured_at_commit") == commit:
                f.write("| %s | %.1f | %.1f | %.3f |\n" % (d["config"]["workload"][:70], d["ms_per_step"], d["forward"]["ms_per_step"],
                                                          d["roofline"]["frac"]))
print("c2: train %.2f ms (%.3f G/s)  fwd %.2f ms (%.3f G/s)  aggregator %.3f ms frac %.3f" % (
    b["ms_per_step"], b["value"] / 1e9, b["forward"]["ms_per_step"], b["forward"]["value"] / 1e9, b["roofline"]["avg_launch_ms"],
    b["roofline"]["frac"]))
