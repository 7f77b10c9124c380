#!/bin/bash
# round 6, first call: whole GPU suite + smoke on the re-recorded traces; default bench line (new contract line)
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/r6a; mkdir -p $O
( time timeout 1500 python -m pytest tests -q -m gpu ) > $O/pytest.txt 2>&1
tail -8 $O/pytest.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3
( time timeout 1200 python bench.py ) > $O/bench_default.out 2> $O/bench_default.err; echo "bench rc=$?"
tail -n 1 $O/bench_default.out
