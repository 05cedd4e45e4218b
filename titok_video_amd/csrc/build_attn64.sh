#!/bin/bash
# ttv_attn64.hip is built in two steps: its kernels own accumulation registers a0 .. a95 through inline asm and need the other 160 of
# a wave's 256 registers as VGPRs.  hipcc splits 128 / 128 by default whenever AGPRs are in use (and then spills); the backend honours
# the function attribute "amdgpu-agpr-alloc"="96", which has no source-level spelling - so the device IR is emitted, the attribute is
# added to the kernels' attribute group, and the device object, its fat binary and the host object are produced by the same tools
# hipcc itself runs (hipcc -### shows the sequence).       usage: build_attn64.sh <out.o> [extra -D flags]
set -e
cd "$(dirname "$0")"
OUT=$1; shift
LLVM=/opt/rocm/lib/llvm/bin
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=on -fno-slp-vectorize -Wall -Wno-unused-function"
T=build/attn64_$$
hipcc $FLAGS "$@" --cuda-device-only -emit-llvm -S ttv_attn64.hip -o $T.ll 2>/dev/null
grep -q '^attributes #0 = { .*"amdgpu-waves-per-eu"="2"' $T.ll || { echo "build_attn64.sh: kernel attribute group not found"; exit 1; }
sed -i 's/^attributes #0 = { /attributes #0 = { "amdgpu-agpr-alloc"="96" /' $T.ll
$LLVM/clang -x ir $T.ll -target amdgcn-amd-amdhsa -mcpu=gfx950 -O3 -fPIC -c -o $T.dev.o -Rpass-analysis=kernel-resource-usage 2> $T.remarks
if grep -q "ScratchSize \[bytes/lane\]: [1-9]" $T.remarks; then echo "build_attn64.sh: WARNING scratch in use"; grep -i "VGPRs\|Scratch" $T.remarks; fi
$LLVM/lld -flavor gnu -m elf64_amdgpu --no-undefined -shared -o $T.out $T.dev.o
$LLVM/clang-offload-bundler -type=o -bundle-align=4096 -targets=host-x86_64-unknown-linux-gnu,hipv4-amdgcn-amd-amdhsa--gfx950 -input=/dev/null -input=$T.out -output=$T.hipfb
hipcc $FLAGS "$@" --cuda-host-only -Xclang -fcuda-include-gpubinary -Xclang $T.hipfb -c ttv_attn64.hip -o $OUT
grep -h "VGPRs:\|AGPRs:\|ScratchSize" $T.remarks | sort | uniq -c
rm -f $T.ll $T.dev.o $T.out $T.hipfb $T.remarks
