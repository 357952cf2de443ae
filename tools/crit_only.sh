#!/bin/bash
# fast iteration on the critical launch alone: compiles ONE instantiation of dec_crit_x3_kernel (device code only) and prints
# its resource usage.  tools/crit_only.sh [-DAAE_CRIT_J4=1 ...]
cd "$(dirname "$0")/.." && mkdir -p build_ab
cat > build_ab/crit_only.hip <<'EOT'
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
#include "../aae-recommender_amd/csrc/gemm_f32.h"
#include "../aae-recommender_amd/csrc/kernels.h"
#include "../aae-recommender_amd/csrc/dec_fused.h"
#include "../aae-recommender_amd/csrc/dec_fused_bf16.h"
#include "../aae-recommender_amd/csrc/dec_crit_x3.h"
template __global__ void aae::dec_crit_x3_kernel<13, false, false, false>(aae::DecFusedArgs);
EOT
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -std=c++17 --cuda-device-only -S -o build_ab/crit_only.s build_ab/crit_only.hip \
   -Rpass-analysis=kernel-resource-usage "$@" 2> build_ab/crit_only.log || { grep -A5 error build_ab/crit_only.log | head -40; exit 1; }
python tools/kernel_usage.py build_ab/crit_only.log dec_crit_x3
