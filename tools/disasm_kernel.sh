#!/bin/bash
# tools/disasm_kernel.sh <lib.so> <mangled-name-substring> -> ISA of the first matching kernel on stdout
set -e
lib=$(realpath $1); pat=$2
tmp=$(mktemp -d); cd $tmp
/opt/rocm/lib/llvm/bin/clang-offload-bundler --list --type=o --input=$lib > /dev/null 2>&1 || true
# the code object is an ELF embedded in .hip_fatbin: extract with roc-obj tools when present, else objcopy
if command -v roc-obj-ls >/dev/null 2>&1; then
  roc-obj-ls $lib | grep gfx950 | awk '{print $NF}' | head -1 > uri.txt
  roc-obj-extract "$(cat uri.txt)" -o co > /dev/null 2>&1 || true
fi
f=$(ls co* 2>/dev/null | head -1)
if [ -z "$f" ]; then
  objcopy -O binary --only-section=.hip_fatbin $lib fat.bin
  /opt/rocm/lib/llvm/bin/clang-offload-bundler --unbundle --type=o --input=fat.bin --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --output=co.elf
  f=co.elf
fi
/opt/rocm/lib/llvm/bin/llvm-objdump -d --mcpu=gfx950 $f | awk -v pat="$pat" '/^[0-9a-f]+ <.*>:/{on=index($0,pat)>0} on{print}'
