#!/bin/bash
# tools/tsan_pool_stress.sh -- ThreadSanitizer run of the host worker pool (host_parallel.hpp) under contention: concurrent
# octree builds from six threads (tests/cpp/pool_stress.cpp).  No GPU needed; prints "done <signature>" and no race report.
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=/tmp/msmhip_tsan
mkdir -p $OUT
cd $ROOT/newmsm_amd/csrc
for f in host_mesh.cpp octree.cpp; do
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -x hip -O1 -g -std=c++17 -fPIC -ffp-contract=off -fsanitize=thread -c $f -o $OUT/${f%.cpp}.o
done
/opt/rocm/lib/llvm/bin/clang++ -fsanitize=thread -O1 -g -std=c++17 -I $ROOT/include -c $ROOT/tests/cpp/pool_stress.cpp -o $OUT/pool_stress.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -fsanitize=thread $OUT/pool_stress.o $OUT/host_mesh.o $OUT/octree.o -o $OUT/pool_stress
MSMHIP_HOST_THREADS=${MSMHIP_HOST_THREADS:-4} $OUT/pool_stress
