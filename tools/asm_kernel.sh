#!/bin/bash
# tools/asm_kernel.sh OUT [extra hipcc flags]: device assembly of chain_kernels.hip (unit 0) and the headline kernel's body cut out of it
out=$1; shift
mkdir -p build/asm
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-slp-vectorize -fassociative-math -fno-signed-zeros -fno-trapping-math --cuda-device-only -S generalized_rbda_amd/csrc/chain_kernels.hip -o build/asm/$out.s -Rpass-analysis=kernel-resource-usage "$@" 2> build/asm/$out.res
awk '/^_ZN9grbda_hip16aba_chain_kernelIfLi2ELi0E.*:/{p=1} p&&!/^\s*;/&&!/^\s*\.[a-z]/{print} p&&/s_endpgm/{exit}' build/asm/$out.s > build/asm/${out}_k.s
grep -A9 'Name: _ZN9grbda_hip16aba_chain_kernelIfLi2ELi0E' build/asm/$out.res | grep 'SGPRs\|VGPRs\|Scratch\|Occupancy' | sed 's/.*remark: //'
wc -l build/asm/${out}_k.s
