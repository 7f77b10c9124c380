#!/bin/bash
# Turns the rocpd databases merged back by tools/profile_round.sh (gpurun_out/prof/) into the
# committed summaries under profiles/.  Usage: bash tools/make_profiles.sh r01
set -e
R=${1:-r02}
P=gpurun_out/prof
# launches of the dominant kernel in the timed step, from the bench line of the PMC pass itself
export FIRST_LAUNCHES=$(grep -h '^{"metric' $P/bench_FETCH_SIZE.log | python -c "import json,sys; print(int(json.loads(sys.stdin.readline())['roofline']['launches_per_step']))")
python tools/rocpd_pmc_traffic.py $P/bench_FETCH_SIZE/bench_results.db $P/bench_WRITE_SIZE/bench_results.db 'gemm_f64_kernel<false, false, 4>' profiles/${R}_pmc_traffic.json > /dev/null
python - profiles/${R}_pmc_traffic.json <<'PY'
import json, sys
rec = json.load(open(sys.argv[1]))
rec['schedule_note'] = ('--pmc serialises kernels, so the counter passes run with DFH_CHOL_LR=0: the trailing updates that the default '
                        'schedule gives to gemm_f64_la_kernel are launches of this kernel there')
rec['scaling'] = 'strong'      # bench.py's default: the whole of config 4 per step (bench.pmc_traffic matches on it)
json.dump(rec, open(sys.argv[1], 'w'), indent=1)
PY
{
echo "rocprofv3 --kernel-trace --stats -- python bench.py --no-cpu-baseline --no-extras --steps 2 --warmup 1   (MI355X, tools/profile_round.sh; rocpd database summarised by tools/rocpd_stats.py)"
echo "4 steps in the trace: 1 warm-up + 2 timed + 1 untimed section-timing step (each: the n = 16384 fit + all 2 097 152 candidates of config 4).  bench.py's own JSON line from this same run:"
grep -h '^{"metric' $P/trace.log
echo
python tools/rocpd_stats.py $P/trace/bench_results.db
echo
echo "The dominant kernel step by step (its first 4 x launches_per_step dispatches; steps 1 and 2 are the timed ones; compare roofline.avg_launch_us_incl_overlap of the bench line above):"
TRACE_LAUNCHES=$(grep -h '^{"metric' $P/trace.log | python -c "import json,sys; print(int(json.loads(sys.stdin.readline())['roofline']['launches_per_step']))")
python - $P/trace/bench_results.db $TRACE_LAUNCHES <<'PY'
import sqlite3, sys
c = sqlite3.connect(sys.argv[1]); n = int(sys.argv[2])
d = [r[0] for r in c.execute("select (end-start)/1e3 from kernels where instr(name,'gemm_f64_kernel<false, false, 4>')>0 order by start")]
# (the look-ahead / conditional variants are kernels of their own, gemm_f64_la_kernel / gemm_f64_cond_kernel)
for s in range(4):
  seg = d[s * n:(s + 1) * n]
  print('  step %d: %d launches, avg %.1f us, total %.1f ms' % (s, len(seg), sum(seg) / len(seg), sum(seg) / 1e3))
PY
echo
echo "PMC passes over the same command (DFH_CHOL_LR=0 rocprofv3 --kernel-trace --pmc FETCH_SIZE | WRITE_SIZE -- python bench.py --no-cpu-baseline --no-extras --steps 1 --warmup 0; kernels are serialised under --pmc, so the schedule that hands tiles over between concurrent kernels is off in these passes):"
python tools/rocpd_pmc_traffic.py $P/bench_FETCH_SIZE/bench_results.db $P/bench_WRITE_SIZE/bench_results.db
} > profiles/${R}_bench_kernel_stats.txt
{
echo "rocprofv3 --kernel-trace --pmc <set> -- python tools/pmc_workload.py   (MI355X, tools/profile_round.sh; one --pmc set per pass)"
echo "pass 1: SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVES SQ_INSTS_VALU_MFMA_MOPS_F64 ; pass 2: FETCH_SIZE ; pass 3: WRITE_SIZE ; pass 4: SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS"
echo "GRBM_GUI_ACTIVE is summed over the 8 XCDs (divide by 8 for cycles); MfmaUtil = MFMA_BUSY / (GUI_ACTIVE/8 * 1024 SIMDs); executed TF/s = MOPS_F64*512 flop / duration"
echo "workload: 3x kernel matrix SE n=16384 d=32 + 2x Matern-2.5 (kernmat_sym_kernel) | 3x cross matrix SE 32768x16384 d=32 + 3x Matern-2.5 65536x4096 d=6 (kernmat_strip_kernel) | 3x SYRK n=15872 k=512 lower | 2x GEMM 8192^3 | one n=4096 factorisation | 2x calibration GEMM 32768x128x16384 (A = 4.295 GB read once + B 16.8 MB per XCD) | round 3: 3x symmetric additive Gram n=4096 d=100 20x5 (kernmat_symmulti_kernel, 134 MB out) | fit n=4096 d=6 Matern-2.5 + 2x EI over 65536 candidates (kernmat_strip_kernel<..,true> with the mean fused in: 2.147 GB out, + k_mu_finish, posterior TRSM)"
echo "FETCH_SIZE calibration: calibration GEMM expects 4.295 + 8*0.0168 = 4.429 GB  => factor 2 for this kernel's 16 B/lane loads (see the last two gemm dispatches of pass 2)"
echo "WRITE_SIZE calibration: kernmat writes 16384^2*8 = 2.147 GB (cross matrices: 4.295 GB and 2.147 GB); SYRK writes the 128-tiles of the lower triangle, 1.016 GB; 8192^3 writes 0.537 GB"
echo
for p in SQ_VALU_MFMA_BUSY_CYCLES FETCH_SIZE WRITE_SIZE SQ_LDS_BANK_CONFLICT; do echo "== pass $p"; python tools/rocpd_pmc_dispatch.py $P/wl_$p/wl_results.db kernmat gemm_f64 k_pack k_mu_finish panel_fused diag_step; echo; done
} > profiles/${R}_pmc_summary.txt
echo "profiles/${R}_* written"
