#!/bin/bash
# Compile one .hip file to a host object through textual ISA, with tools/isa_vccfix.py between the compiler and the assembler:
#   tools/hipcc_isa.sh ARCH SRC.hip OUT.o [compile flags...]
# The steps are the ones `hipcc -c` runs internally (hipcc -### shows them): device compile, link to a code object, bundle, host
# compile with the bundle embedded.  Only the device compile is stopped at assembly so the pass can run on it.
set -e
ARCH=$1; SRC=$2; OUT=$3; shift 3
ROCM=${ROCM_PATH:-/opt/rocm}
LLVM=$ROCM/lib/llvm/bin
HERE=$(dirname "$(readlink -f "$0")")
T=${OUT%.o}
$ROCM/bin/hipcc --offload-arch=$ARCH "$@" --cuda-device-only -S "$SRC" -o "$T.dev.s" 2> >(grep -v "argument unused during compilation" >&2)
python3 "$HERE/isa_vccfix.py" "$T.dev.s" "$T.fix.s"
$LLVM/clang -x assembler -target amdgcn-amd-amdhsa -mcpu=$ARCH -c "$T.fix.s" -o "$T.dev.o"
$LLVM/lld -flavor gnu -m elf64_amdgpu --no-undefined -shared "$T.dev.o" -o "$T.co"
$LLVM/clang-offload-bundler -type=o -bundle-align=4096 -targets=host-x86_64-unknown-linux-gnu,hipv4-amdgcn-amd-amdhsa--$ARCH -input=/dev/null -input="$T.co" -output="$T.hipfb"
$ROCM/bin/hipcc --offload-arch=$ARCH "$@" --cuda-host-only -Xclang -fcuda-include-gpubinary -Xclang "$T.hipfb" -c "$SRC" -o "$OUT"
