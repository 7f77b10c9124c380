#!/bin/bash
# round 6, closing measurements: whole GPU suite + smoke; the default bench line; rocprofv3 kernel trace + PMC traffic
# passes of bench.py (tools/profile_round.sh bench-only)
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/r6n; mkdir -p $O
( time timeout 1800 python -m pytest tests -q -m gpu ) > $O/pytest.txt 2>&1
tail -8 $O/pytest.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3
( time timeout 1500 python bench.py ) > $O/bench_default.out 2> $O/bench_default.err; echo "bench rc=$?"
tail -n 1 $O/bench_default.out | cut -c1-3200
bash tools/profile_round.sh bench-only > $O/profile_round.log 2>&1; echo "profile rc=$?"
ls -la gpurun_out/prof/*/ | head -20
