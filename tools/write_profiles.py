#!/usr/bin/env python3
"""Copy measurement sets taken by tools/measure_workload.sh (gpurun_out/w_<workload>_*) into profiles/<round>_*:
the bench line, the rocprofv3 kernel statistics, the PMC traffic per launch and the SQ counters, per workload, each
stamped with the commit it was measured at, plus one markdown table per workload that puts them side by side.

    python tools/write_profiles.py r03 c2 c4 c5 [c3 ...]
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
sys.path.insert(0, os.path.join(ROOT, "tools"))


def counters(dirname, names):
    """{kernel: {counter: mean per launch}} for the mpnn kernels of one rocprofv3 --pmc pass."""
    fs = glob.glob(dirname + "/**/*counter_collection.csv", recursive=True)
    if not fs:
        return {}
    acc = {}
    for r in csv.DictReader(open(fs[0])):
        if r["Counter_Name"] in names and "mpnn::" in r["Kernel_Name"]:
            k = r["Kernel_Name"].split("(")[0].replace("void ", "")
            acc.setdefault(k, {}).setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
    return {k: {c: sum(v) / len(v) for c, v in d.items()} for k, d in acc.items()}


def main():
    rnd, workloads = sys.argv[1], sys.argv[2:]
    for w in workloads:
        pre = os.path.join(G, "w_%s" % w)
        commit = open(pre + "_commit.txt").read().strip() if os.path.exists(pre + "_commit.txt") else None
        if not commit:
            # the GPU box gets a snapshot of the working tree without .git: the commit is the last local one that touches the
            # code the snapshot ran (run this script before committing further code changes)
            commit = subprocess.run(["git", "log", "-1", "--format=%H", "--", "mpnn_amd", "bench.py", "include"], cwd=ROOT,
                                    capture_output=True, text=True).stdout.strip() or None      # the last commit that touches code
        bench = None
        if os.path.exists(pre + "_bench.log"):
            lines = [x for x in open(pre + "_bench.log") if x.startswith("{")]
            if lines:
                bench = json.loads(lines[-1])
                bench["measured_at_commit"] = commit
                open(os.path.join(P, "%s_bench_%s.json" % (rnd, w)), "w").write(json.dumps(bench) + "\n")
        fetch = counters(pre + "_fetch", ("FETCH_SIZE",))
        write = counters(pre + "_write", ("WRITE_SIZE",))
        pmc = None
        if fetch:
            pmc = {"workload": w, "commit": commit,
                   "source": "rocprofv3 --kernel-trace --pmc FETCH_SIZE | WRITE_SIZE (separate passes) -- python3 bench.py "
                             "--workload %s --steps 2 --warmup 1 --no-cpu --no-side" % w,
                   "corrections": "KiB -> bytes (x1024); FETCH_SIZE x2 (gfx950 counts a 16 B/lane coalesced stream at half)",
                   "kernels": {}}
            for k, d in fetch.items():
                rd = d["FETCH_SIZE"] * 1024 * 2
                wr = write.get(k, {}).get("WRITE_SIZE", 0.0) * 1024
                pmc["kernels"][k] = {"fetch_size_kib_raw": d["FETCH_SIZE"], "write_size_kib_raw": write.get(k, {}).get("WRITE_SIZE"),
                                     "hbm_read_bytes": rd, "hbm_write_bytes": wr, "hbm_bytes": rd + wr}
            json.dump(pmc, open(os.path.join(P, "%s_pmc_%s.json" % (rnd, w)), "w"), indent=1)
        sq = None
        if glob.glob(pre + "_sq/**/*counter_collection.csv", recursive=True):
            subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "sq_summary.py"), pre + "_sq",
                                   os.path.join(P, "%s_sq_%s.json" % (rnd, w))], stdout=subprocess.DEVNULL)
            sq = json.load(open(os.path.join(P, "%s_sq_%s.json" % (rnd, w))))["kernels"]
        stats = glob.glob(pre + "_prof/**/*kernel_stats.csv", recursive=True)
        if not stats:
            continue
        open(os.path.join(P, "%s_%s_kernel_stats.csv" % (rnd, w)), "w").write(open(stats[0]).read())
        rows = list(csv.DictReader(open(stats[0])))
        tot = sum(float(r["TotalDurationNs"]) for r in rows)
        with open(os.path.join(P, "%s_%s_kernel_stats.md" % (rnd, w)), "w") as f:
            f.write("# %s: `rocprofv3 --kernel-trace --stats -- python3 bench.py --workload %s --no-cpu --no-side` (commit %s)\n\n"
                    "Forward-only passes, training passes and the per-kernel event pass of one bench run.  HBM bytes per launch from the "
                    "FETCH_SIZE / WRITE_SIZE passes of the same command (`%s_pmc_%s.json`; KiB -> bytes, FETCH x2 for gfx950's "
                    "half-counted wide reads); SQ columns from one SQ counter pass (`%s_sq_%s.json`): fraction of wave cycles parked "
                    "on s_waitcnt / barrier, and matrix-pipe busy fraction of the kernel's duration per SIMD.\n\n"
                    "| kernel | calls | avg ms | %% of GPU time | HBM read GB | HBM write GB | TB/s moved | parked | MFMA busy |\n"
                    "|---|---|---|---|---|---|---|---|---|\n" % (rnd, w, (commit or "?")[:12], rnd, w, rnd, w))
            for r in rows[:16]:
                name = r["Name"].split("(")[0].replace("void ", "")
                ms = float(r["AverageNs"]) / 1e6
                pk = (pmc or {}).get("kernels", {}).get(name)
                sk = (sq or {}).get(name)
                f.write("| `%s` | %s | %.3f | %.1f | %s | %s | %s | %s | %s |\n" % (
                    name[:80], r["Calls"], ms, 100 * float(r["TotalDurationNs"]) / tot,
                    "%.2f" % (pk["hbm_read_bytes"] / 1e9) if pk else "", "%.2f" % (pk["hbm_write_bytes"] / 1e9) if pk else "",
                    "%.2f" % (pk["hbm_bytes"] / ms / 1e9) if pk else "",
                    "%.2f" % sk["wave_parked_frac"] if sk else "",
                    ("%.2f" % sk["mfma_busy_frac"]) if sk and sk["mfma_busy_frac"] is not None else ""))
            if bench:
                f.write("\nUn-profiled `python bench.py --workload %s` at the same commit (`%s_bench_%s.json`): training step %.2f ms = "
                        "%.3f G edges/s, forward pass %.2f ms.\n" % (w, rnd, w, bench["ms_per_step"], bench["value"] / 1e9,
                                                                     bench["forward"]["ms_per_step"]))
        print("%s: wrote profiles/%s_%s_kernel_stats.md%s" % (w, rnd, w, "  train %.2f ms" % bench["ms_per_step"] if bench else ""))


if __name__ == "__main__":
    main()
