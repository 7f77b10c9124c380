#!/bin/bash
# round 5, closing measurements: the default bench line; rocprofv3 kernel trace + PMC traffic passes of bench.py
# (tools/profile_round.sh bench-only); kernel traces and an MFMA-utilisation pass of the tuning objective after the round.
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/r5o; mkdir -p $O
R=$GRAFT_REPO_ROOT
( time timeout 1200 python bench.py ) > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?"
bash tools/profile_round.sh bench-only > $O/profile_round.log 2>&1; echo "profile rc=$?"
( cd /tmp && export TMPDIR=/tmp
  for cfg in "1000 64 5" "1000 2048 2" "200 4096 2" "2000 512 2"; do
    set -- $cfg
    timeout 300 rocprofv3 --kernel-trace --stats -d $O/lml_n$1_nb$2 -o t -- python $R/tools/prof_lml.py $1 $2 $3 > $O/lml_n$1_nb$2.log 2>&1
  done
  timeout 300 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVES SQ_INSTS_VALU_MFMA_MOPS_F64 -d $O/lml_pmc -o t -- python $R/tools/prof_lml.py 1000 2048 1 > $O/lml_pmc.log 2>&1 )
find $O $R/gpurun_out/prof -name '*.db' -size +30M -delete
grep -h "ms per call" $O/lml_n*.log; tail -c 1500 $O/bench_default.json; ls -la $R/gpurun_out/prof/*/ | head -20
