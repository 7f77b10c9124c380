#!/bin/bash
# round 5: does the posterior chunk size (rows per GEMM launch) change the step time?  (GEMM re-reads: 2.8x algorithmic
# with 2 097 152-row launches)
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/r5r; mkdir -p $O
for gib in 32 8 2 1; do
  DFH_CHUNK_GIB=$gib timeout 300 python bench.py --no-cpu-baseline --no-extras --steps 1 --warmup 1 2> /dev/null | grep '^{"metric' > $O/bench_chunk_$gib.json
  python - $O/bench_chunk_$gib.json $gib <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read())
r = d['roofline']
print('DFH_CHUNK_GIB=%s: ms_per_step %.1f  gemm frac %.4f  launches/step %d  avg_launch_us %.0f  trsm_ms %.1f ts_ms %.1f cross_ms %.1f' % (
  sys.argv[2], d['ms_per_step'], r['frac'], r['launches_per_step'], r['avg_launch_us'], d['trsm_ms'], d['ts_ms'], d['cross_ms']))
PY
done
