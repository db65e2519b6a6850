#!/bin/bash
# One workload's full measurement set on the GPU box (run through gpurun): the bench line, rocprofv3 kernel statistics of
# the same command, the HBM-traffic PMC passes (FETCH_SIZE and WRITE_SIZE in separate runs) and one SQ counter pass.
# Everything lands in gpurun_out/w_<workload>_*; afterwards, on the build machine,
#     python tools/write_profiles.py r04 <workloads...>
# copies the summaries into profiles/r04_*.
#   usage: bash tools/measure_workload.sh <workload> [bench|stats|pmc|sq ...]   (default: all four)
set -e
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
w="$1"; shift
WHAT="$@"; [ -z "$WHAT" ] && WHAT="bench stats pmc sq"
S=3; case "$w" in c2|c1|tiny) S=10;; c3|c4) S=5;; esac
P="gpurun_out/w_${w}"
git rev-parse HEAD > ${P}_commit.txt 2>/dev/null || true
for what in $WHAT; do
  case $what in
    bench) timeout -k 10 600 python bench.py --workload $w --steps $S --warmup 3 > ${P}_bench.log 2> ${P}_bench.err ;;
    stats) rm -rf ${P}_prof
           timeout -k 10 400 rocprofv3 --kernel-trace --stats -d ${P}_prof -o $w --output-format csv -- python3 bench.py --workload $w --steps $S --warmup 2 --no-cpu --no-side > ${P}_prof.log 2>&1 ;;
    pmc)   rm -rf ${P}_fetch ${P}_write
           timeout -k 10 400 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d ${P}_fetch -o $w --output-format csv -- python3 bench.py --workload $w --steps 2 --warmup 1 --no-cpu --no-side > ${P}_fetch.log 2>&1
           timeout -k 10 400 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d ${P}_write -o $w --output-format csv -- python3 bench.py --workload $w --steps 2 --warmup 1 --no-cpu --no-side > ${P}_write.log 2>&1 ;;
    sq)    rm -rf ${P}_sq
           timeout -k 10 400 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_VALU_MFMA_MOPS_BF16 GRBM_GUI_ACTIVE -d ${P}_sq -o $w --output-format csv -- python3 bench.py --workload $w --steps 2 --warmup 1 --no-cpu --no-side > ${P}_sq.log 2>&1 ;;
  esac
  echo "$w $what done"
done
[ -f ${P}_bench.log ] && cut -c1-200 ${P}_bench.log || true
