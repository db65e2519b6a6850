#!/bin/bash
# Per-kernel times of one command on the GPU box: bash tools/prof_kernels.sh <tag> <pattern> -- python3 <script> [args]
# (rocprofv3 --kernel-trace --stats into gpurun_out/<tag>/, then the kernel_stats rows matching <pattern>).
tag=$1; pat=$2; shift 3
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d gpurun_out/$tag -o p --output-format csv -- "$@" > gpurun_out/$tag.log 2>&1
echo "rc=$?"
python3 - "$tag" "$pat" <<'PY'
import csv, re, sys
tag, pat = sys.argv[1], sys.argv[2]
rows = list(csv.DictReader(open("gpurun_out/%s/p_kernel_stats.csv" % tag)))
for r in rows:
    if re.search(pat, r["Name"]):
        print("%-70s calls %4s avg %9.1f us  min %9.1f  max %9.1f" % (r["Name"][:70], r["Calls"], float(r["AverageNs"]) / 1e3,
              float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3))
PY
