#!/bin/bash
# instruction mix of one kernel (substring of its demangled name) in mpboot_amd/csrc/_obj/<obj>.o:  tools/isa_counts.sh kernels 'k_snk_scan<20, 6, true, false, true>'
set -e
cd "$(dirname "$0")/.."
LLVM=/opt/rocm/lib/llvm/bin
TMP=$(mktemp -d)
$LLVM/llvm-objcopy --dump-section .hip_fatbin=$TMP/a.fb mpboot_amd/csrc/_obj/$1.o
$LLVM/clang-offload-bundler --type=o --input=$TMP/a.fb --unbundle --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --output=$TMP/a.co
$LLVM/llvm-objdump -d $TMP/a.co | c++filt | awk -v pat="$2" '
  /^[0-9a-f]+ <.*>:$/ { on = index($0, pat) > 0; if (on) print $0 }
  on && NF > 1 && $1 !~ /^[0-9a-f]+$/ { n[$1]++; tot++ }
  END { for (k in n) print n[k], k; print tot, "TOTAL" }' | sort -rn | head -${3:-40}
rm -rf $TMP
