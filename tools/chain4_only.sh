#!/bin/bash
# fast iteration on the 4-row chain kernel alone: compiles ONE instantiation of chain4_kernel (device code only) and prints
# its resource usage.  tools/chain4_only.sh [-D...]
cd "$(dirname "$0")/.." && mkdir -p build_ab
cat > build_ab/chain4_only.hip <<'EOT'
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include <string>
#include <vector>
#include "../include/aaerec_hip.h"
#include "../aae-recommender_amd/csrc/chain4.h"
template __global__ void aae::chain4_kernel<false, false>(aae::ChainProgram);
EOT
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -std=c++17 --cuda-device-only -S -o build_ab/chain4_only.s build_ab/chain4_only.hip \
   -Rpass-analysis=kernel-resource-usage "$@" 2> build_ab/chain4_only.log || { grep -A5 error build_ab/chain4_only.log | head -40; exit 1; }
python tools/kernel_usage.py build_ab/chain4_only.log chain4_kernel
