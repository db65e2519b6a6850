#!/bin/bash
# Timing of the H = 128 GRU backward (tools/bench_gru_bwd.py under rocprofv3): the product build, the round-3 form
# (MPNN_GRU_BWD=pieces), and builds with experiment macros (wrong results, timing only; each in its own library directory,
# mpnn_amd/build.py), one run per flag set:   bash tools/abl_rc.sh [flagset ...]     e.g.  -DMPNN_ABL_RC_HOTROWS
cd "$GRAFT_REPO_ROOT" || exit 1
H=${RC_H:-128}
bash tools/prof_kernels.sh abl_rc_base "gru_(rc|bwd|gate)" -- python3 tools/bench_gru_bwd.py $H time
MPNN_GRU_BWD=pieces bash tools/prof_kernels.sh abl_rc_pieces "gru_(rc|bwd|gate)" -- python3 tools/bench_gru_bwd.py $H time
i=0
for f in "$@"; do
  i=$((i+1))
  echo "== $f"
  MPNN_EXTRA_HIPCC_FLAGS="$f" MPNN_BUILD_JOBS=16 python3 -m mpnn_amd.build > gpurun_out/abl_build.log 2>&1 || { tail -5 gpurun_out/abl_build.log; exit 1; }
  MPNN_EXTRA_HIPCC_FLAGS="$f" bash tools/prof_kernels.sh abl_rc_$i "gru_(rc|bwd|gate)" -- python3 tools/bench_gru_bwd.py $H time
done
