#!/bin/bash
# The scalar-wave-index anomaly, proven by a one-instruction patch of the compiler's own listing (docs/NOTES_r05.md section 2).
# Builds, from ONE hipcc -save-temps compilation of chol.hip with `const int w = readfirstlane(tid >> 6)` at the scope of
# panel_fused_kernel<true> (diagnostics build):
#   tools/_exp/libdfhip_dbg_ctrl.so  the device listing re-assembled as it is (the object must equal the compiler's, checked)
#   tools/_exp/libdfhip_dbg_fix.so   the listing + `v_accvgpr_mov_b32 a195, a191` in the strip-0 leaf of the block select
# then `gpurun -- 'bash tools/r5_run20.sh'` compares both against LAPACK.  Run in the build container from the repo root.
set -e
R=$(pwd); W=/tmp/nanv2; B=/opt/rocm/lib/llvm/bin; mkdir -p $W/patched $R/tools/_exp
python -m dragonfly_amd.build --debug-hooks > /dev/null
cp $R/dragonfly_amd/csrc/common.h $W/ && sed -i "s#\"../../include/dfhip.h\"#\"$R/include/dfhip.h\"#" $W/common.h
python3 - "$R" "$W" <<'PY'
import sys
R, W = sys.argv[1], sys.argv[2]
s = open(R + '/dragonfly_amd/csrc/chol.hip').read()
old = "  const int w = tid >> 6;\n  const int kq = lane >> 4, l15 = lane & 15;\n  const int g = blockIdx.x + a.g0;"
assert s.count(old) == 1
open(W + '/chol_v.hip', 'w').write(s.replace(old, old.replace("tid >> 6;", "__builtin_amdgcn_readfirstlane(tid >> 6);", 1)))
PY
( cd $W && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -DDFH_DEBUG_HOOKS -save-temps -c chol_v.hip -o chol_bad.o )
DEV=$W/chol_v-hip-amdgcn-amd-amdhsa-gfx950.s; HOST=$W/chol_v-host-x86_64-unknown-linux-gnu.s
python tools/isa_audit.py $DEV || true          # expected: a191 in panel_fused_kernel<true>
python3 - $DEV $W/patched/dev.s <<'PY'
import sys
L = open(sys.argv[1]).read().split('\n')
idx = [i for i, l in enumerate(L) if 'scratch_load_dwordx3 a[192:194], off, off offset:384' in l]
assert len(idx) == 1, 'the allocation differs from the one docs/NOTES_r05.md describes: %r' % idx
j = next(k for k in range(idx[0], idx[0] + 12) if 'v_accvgpr_mov_b32 a196, a190' in L[k])
L.insert(j + 1, '\tv_accvgpr_mov_b32 a195, a191            ; PATCH: the piece the reload left out')
open(sys.argv[2], 'w').write('\n'.join(L))
PY
mk() {   # $1 = device listing, $2 = tag: assemble, link the code object, bundle, embed in the host listing, assemble that
  cd $W/patched
  $B/clang -cc1as -triple amdgcn-amd-amdhsa -filetype obj -main-file-name chol_v.hip -target-cpu gfx950 -mrelocation-model pic -o $2.dev.o $1
  $B/lld -flavor gnu -m elf64_amdgpu --no-undefined -shared -plugin-opt=-amdgpu-internalize-symbols --lto-partitions=8 \
         -plugin-opt=mcpu=gfx950 -plugin-opt=O3 --lto-CGO3 --whole-archive -o $2.out $2.dev.o --no-whole-archive
  $B/clang-offload-bundler -type=o -bundle-align=4096 -targets=host-x86_64-unknown-linux-gnu,hipv4-amdgcn-amd-amdhsa--gfx950 \
         -input=/dev/null -input=$2.out -output=$2.hipfb
  python3 - $HOST $2 <<'PY'
import os, sys
host, tag = sys.argv[1], sys.argv[2]
L = open(host, errors='surrogateescape').read().split('\n')
i = next(k for k, l in enumerate(L) if l.startswith('\t.asciz\t"__CLANG_OFFLOAD_BUNDLE__'))
L[i] = '\t.incbin\t"%s/%s.hipfb"' % (os.getcwd(), tag)
assert L[i + 1].startswith('\t.size\t.L__unnamed_')
L[i + 1] = L[i + 1].split(',')[0] + ', %d' % os.path.getsize(tag + '.hipfb')
open(tag + '.host.s', 'w', errors='surrogateescape').write('\n'.join(L))
PY
  $B/clang -cc1as -triple x86_64-unknown-linux-gnu -filetype obj -main-file-name chol_v.hip -target-cpu x86-64 -mrelocation-model pic -o chol_$2.o $2.host.s
  cd $R
}
mk $DEV ctrl; mk $W/patched/dev.s fix
cmp $W/patched/chol_ctrl.o $W/chol_bad.o && echo "ctrl == the compiler's own object"
objs=$(ls $R/dragonfly_amd/csrc/_obj_dbg/*.o | grep -v "/chol.o")
for v in ctrl fix; do /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $R/tools/_exp/libdfhip_dbg_$v.so $objs $W/patched/chol_$v.o; done
python tools/isa_audit.py $R/tools/_exp/libdfhip_dbg_ctrl.so $R/tools/_exp/libdfhip_dbg_fix.so || true
