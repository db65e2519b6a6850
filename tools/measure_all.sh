#!/bin/bash
# One measurement set on the GPU box (run through gpurun): GPU tests, the c2 bench line, the rocprofv3 kernel statistics and
# the PMC passes of the same command, then the other workloads.  Everything lands in gpurun_out/m_*; afterwards, on the
# build machine,   python tools/refresh_profiles.py   copies the summaries into profiles/r02_* (this script and that one
# are the only writers of profiles/r02_*).
#   usage: bash tools/measure_all.sh [tests|notests] [workloads...]
set -e
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
W="${@:2}"; [ -z "$W" ] && W="c3 c4 c5 c1"
if [ "$1" != "notests" ]; then
    timeout -k 10 900 python -m pytest tests -m gpu -q -x > gpurun_out/m_tests.log 2>&1 || { tail -20 gpurun_out/m_tests.log; exit 1; }
    tail -1 gpurun_out/m_tests.log
fi
git rev-parse HEAD > gpurun_out/m_commit.txt 2>/dev/null || true
timeout -k 10 300 python bench.py > gpurun_out/m_bench_c2.log 2> gpurun_out/m_bench_c2.err
rm -rf gpurun_out/m_prof gpurun_out/m_fetch gpurun_out/m_write gpurun_out/m_sq
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d gpurun_out/m_prof -o c2 --output-format csv -- python3 bench.py --no-cpu > gpurun_out/m_prof.log 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d gpurun_out/m_fetch -o c2 --output-format csv -- python3 bench.py --steps 2 --warmup 1 --no-cpu > gpurun_out/m_fetch.log 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d gpurun_out/m_write -o c2 --output-format csv -- python3 bench.py --steps 2 --warmup 1 --no-cpu > gpurun_out/m_write.log 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_VALU_MFMA_MOPS_BF16 GRBM_GUI_ACTIVE -d gpurun_out/m_sq -o c2 --output-format csv -- python3 bench.py --steps 2 --warmup 1 --no-cpu > gpurun_out/m_sq.log 2>&1
for w in $W; do
    timeout -k 10 400 python bench.py --workload $w --warmup 3 > gpurun_out/m_bench_$w.log 2>gpurun_out/m_bench_$w.err
    echo done $w
done
# BASELINE configs[3] as a strong-scaling run at one rank: 1 M molecules in 8 micro-batches per step
timeout -k 10 700 python bench.py --workload c4 --scaling strong --steps 3 --warmup 1 --no-cpu > gpurun_out/m_bench_c4strong.log 2>gpurun_out/m_bench_c4strong.err
cut -c1-260 gpurun_out/m_bench_c2.log
