export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-.}"
rm -rf gpurun_out/mt_sq gpurun_out/mt_sq2
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE -d gpurun_out/mt_sq -o mt --output-format csv -- python3 tools/bench_message_tile.py > gpurun_out/mt_sq.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_MISC GRBM_GUI_ACTIVE -d gpurun_out/mt_sq2 -o mt --output-format csv -- python3 tools/bench_message_tile.py > gpurun_out/mt_sq2.log 2>&1
python3 - <<'PY'
import csv, glob
for d in ("gpurun_out/mt_sq", "gpurun_out/mt_sq2"):
    fs = glob.glob(d + "/**/*counter_collection.csv", recursive=True)
    if not fs:
        print("no csv in", d); continue
    acc = {}
    for r in csv.DictReader(open(fs[0])):
        if "message_sum_tile" not in r["Kernel_Name"]:
            continue
        acc.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
    for k, v in sorted(acc.items()):
        print("%-28s %.4g  (n=%d)" % (k, sum(v) / len(v), len(v)))
PY
