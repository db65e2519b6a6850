#!/bin/bash
# Timing of the wide GRU kernels (tools/bench_gru_bwd.py under rocprofv3) in the product build and in builds with extra
# compiler flags, one run per flag set:
#   bash tools/abl_gru.sh <H> [flagset ...]        e.g.  bash tools/abl_gru.sh 128 -DMPNN_DX_NW=4 -DMPNN_ABL_HOT_ROWS
# Experiment macros (wrong results, timing only).  Forward kernel (gru_split.hip): MPNN_ABL_HOT_ROWS = row operands read
# from one L2-resident tile, MPNN_ABL_NO_WCOPY = weight chunks never refreshed, MPNN_ABL_NO_EPI_MATH = gate nonlinearities
# left out.  dm | dh kernel (gru_bwd128_f16.hip): MPNN_ABL_DX_NOSCALE = the rows' scales neither fetched nor undone.  (The
# HOT_ROWS / NO_WCOPY pair was first used on the one-chunk-ahead dm | dh kernel that round 3 replaced: DESIGN 3c.)
# Kernel times vary by up to 15 % between boxes: compare runs of ONE call only.
# A build with MPNN_EXTRA_HIPCC_FLAGS set lands in mpnn_amd/lib/variant_<hash of the flags>/ and is loaded from there by
# processes that carry the same variable (mpnn_amd/build.py): the product library is never replaced by an experiment.
cd "$GRAFT_REPO_ROOT" || exit 1
H=$1; shift
bash tools/prof_kernels.sh abl_base_$H "gru_(update|bwd_d|gate)" -- python3 tools/bench_gru_bwd.py $H time
i=0
for f in "$@"; do
  i=$((i+1))
  echo "== $f"
  MPNN_EXTRA_HIPCC_FLAGS="$f" python3 -m mpnn_amd.build > gpurun_out/abl_build.log 2>&1 || { tail -5 gpurun_out/abl_build.log; exit 1; }
  MPNN_EXTRA_HIPCC_FLAGS="$f" bash tools/prof_kernels.sh abl_${i}_$H "gru_(update|bwd_d|gate)" -- python3 tools/bench_gru_bwd.py $H time
done
