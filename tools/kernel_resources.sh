#!/bin/bash
# Registers, LDS and scratch of every kernel in the built objects (mpboot_amd/csrc/_obj/*.o), from the code objects' metadata;
# with -s also the number of scratch_ instructions per kernel from the disassembly.
#   tools/kernel_resources.sh [-s] [pattern]
set -e
cd "$(dirname "$0")/.."
LLVM=/opt/rocm/lib/llvm/bin
SCR=0
if [ "$1" = "-s" ]; then SCR=1; shift; fi
PAT="${1:-.}"
TMP=$(mktemp -d)
for o in mpboot_amd/csrc/_obj/kernels.o mpboot_amd/csrc/_obj/climb.o mpboot_amd/csrc/_obj/grow.o mpboot_amd/csrc/_obj/ufboot.o mpboot_amd/csrc/_obj/reps.o; do
  [ -f "$o" ] || continue
  b=$(basename "$o" .o)
  $LLVM/llvm-objcopy --dump-section .hip_fatbin=$TMP/$b.fb "$o" 2>/dev/null || continue
  $LLVM/clang-offload-bundler --type=o --input=$TMP/$b.fb --unbundle --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --output=$TMP/$b.co 2>/dev/null || continue
  $LLVM/llvm-readelf --notes "$TMP/$b.co" | python3 -c "
import sys, re
txt = sys.stdin.read()
for blk in txt.split('  - .agpr_count:')[1:]:
    g = lambda k: (re.search(r'\.' + k + r':\s+(\S+)', blk) or [None, '?'])[1]
    print('%-8s vgpr %4s agpr %4s sgpr %4s lds %7s scratch %6s  %s' % ('$b', g('vgpr_count'), blk.split()[0], g('sgpr_count'), g('group_segment_fixed_size'), g('private_segment_fixed_size'), g('name')))
" | c++filt | grep -E "$PAT" || true
  if [ $SCR = 1 ]; then
    $LLVM/llvm-objdump -d "$TMP/$b.co" | awk '/^[0-9a-f]+ <.*>:$/ {name=$2} /scratch_/ {n[name]++} END {for (k in n) print n[k], k}' | tr -d '<>:' | c++filt | sort -rn | grep -E "$PAT" | sed "s/^/  scratch_ instructions: /" || true
  fi
done
rm -rf "$TMP"
