#!/bin/bash
# One measurement set on the GPU box (run through gpurun): GPU tests, the c2 bench line, the rocprofv3 kernel statistics
# and the PMC passes behind profiles/r01_*, then the other workloads.  Afterwards, on the build machine:
#   python tools/refresh_profiles.py gpurun_out/f_bench.log gpurun_out/f_prof gpurun_out/f_fetch gpurun_out/f_write
#   python tools/sq_summary.py gpurun_out/f_sq profiles/r01_sq_c2.json
set -e
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
python -m pytest tests -m gpu -q -x > gpurun_out/f_tests.log 2>&1
tail -1 gpurun_out/f_tests.log
python bench.py > gpurun_out/f_bench.log 2>gpurun_out/f_bench.err
rm -rf gpurun_out/f_prof gpurun_out/f_fetch gpurun_out/f_write gpurun_out/f_sq
rocprofv3 --kernel-trace --stats -d gpurun_out/f_prof -o c2 --output-format csv -- python3 bench.py --no-cpu > gpurun_out/f_prof.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d gpurun_out/f_fetch -o c2 --output-format csv -- python3 bench.py --steps 2 --warmup 1 --no-cpu > gpurun_out/f_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d gpurun_out/f_write -o c2 --output-format csv -- python3 bench.py --steps 2 --warmup 1 --no-cpu > gpurun_out/f_write.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_VALU_MFMA_MOPS_BF16 GRBM_GUI_ACTIVE -d gpurun_out/f_sq -o c2 --output-format csv -- python3 bench.py --steps 2 --warmup 1 --no-cpu > gpurun_out/f_sq.log 2>&1
for w in c3 c3a c4 c1 c5; do
    python bench.py --workload $w --warmup 3 > gpurun_out/f_bench_$w.log 2>gpurun_out/f_bench_$w.err
    echo done $w
done
cut -c1-300 gpurun_out/f_bench.log
