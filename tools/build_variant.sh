#!/bin/bash
# A/B builds of the library: tools/build_variant.sh <name> [extra hipcc flags...] -> aae-recommender_amd/aaerec/libaaerec_hip_<name>.so
# (picked up with AAE_HIP_LIB=<path>; the resource usage of the output-layer kernels goes to build_ab/<name>.usage)
set -e
cd "$(dirname "$0")/.."
name=$1; shift
out=aae-recommender_amd/aaerec/libaaerec_hip_${name}.so
mkdir -p build_ab
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -std=c++17 -fPIC -shared -o $out aae-recommender_amd/csrc/aae_abi.hip \
    -Wl,-rpath,/opt/rocm/lib -Rpass-analysis=kernel-resource-usage "$@" 2> build_ab/${name}.log || { grep -E "error" -A5 build_ab/${name}.log | head -50; exit 1; }
grep -A12 "Function Name: .*\(dec_crit_x3\|rank_x3\)" build_ab/${name}.log | grep -E "Function Name|VGPRs:|Spill|LDS Size|Occupancy" > build_ab/${name}.usage || true
echo built $out
