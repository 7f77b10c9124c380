/*
 * dfhip_debug.h -- diagnostics hooks of libdfhip.so.  NOT part of the product C-ABI (include/dfhip.h):
 * they exist only in libdfhip_dbg.so, built with -DDFH_DEBUG_HOOKS (`python -m dragonfly_amd.build
 * --debug-hooks`, loaded with DFH_LIB=...), are used by tools/dbg_*.py during kernel work, and replace nothing in the
 * reference.
 */
#ifndef DFHIP_DEBUG_H
#define DFHIP_DEBUG_H
#include "dfhip.h"
#ifdef __cplusplus
extern "C" {
#endif
/* Do kernels of two streams of one context run concurrently?  which: 0 = bulk vs main, 1 = main vs
 * panel, 2 = bulk vs panel.  out_ms[4]: big alone, small alone, small-stream completion when both
 * run, packed total/priorities.                                                                */
int dfh_debug_overlap(dfh_ctx* ctx, int which, int n_big, int n_small, double* out_ms);
/* Achievable HBM bandwidth of plain kernels: out[3] = 16-byte fill, hipMemset, copy (TB/s).      */
int dfh_debug_write_bw(dfh_ctx* ctx, double gbytes, double* out);
/* `reps` back-to-back launches of the 64-wide pivot step of the Cholesky factorisation on a
 * synthetic block: ms per launch and in-kernel cycle stamps.                                    */
int dfh_debug_diag_step(dfh_ctx* ctx, int reps, int rows_below, double* ms_per_launch, long long* cycles_out);
/* `reps` launches of the one-launch panel (panel_fused_kernel) on a synthetic SPD 512 x 512 block with
 * `rows_below` rows under it: ms_out[reps] per launch; stamps_out[(8 + ceil(rows_below/64))][64] =
 * s_memrealtime (100 MHz) at the marked points of the last launch (tools/dbg_panel.py decodes them). */
int dfh_debug_panel_stamps(dfh_ctx* ctx, int reps, int rows_below, double* ms_out, long long* stamps_out);
/* input block and published 64 x 64 factor blocks of the last dfh_debug_panel_stamps call (either may be NULL);
 * returns the number of doubles of the input block.                                                        */
int dfh_debug_panel_data(double* A_out, double* Lfac_out);
/* The team form of the one-workgroup tuning objective (lml_team_kernel) stamps its progress into dev_buf
 * ([workgroups][32 block columns][16] int64 of s_memrealtime, zeroed by the caller) from the next launch on;
 * NULL switches the stamps off (tools/dbg_lmlt.py).                                                        */
int dfh_debug_lmlt_stamps(void* dev_buf);
#ifdef __cplusplus
}
#endif
#endif /* DFHIP_DEBUG_H */
