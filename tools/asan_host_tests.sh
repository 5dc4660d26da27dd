#!/bin/bash
# tools/asan_host_tests.sh -- AddressSanitizer run of the library's host code (octree builds on the worker pool, icosphere,
# adjacency, label grids, Monte Carlo optimiser, variance normalisation, ABI checks) on a machine WITHOUT a GPU: the host
# translation units are rebuilt with -fsanitize=address into /tmp/msmhip_asan and the CPU tests that call [host] entry points
# run against that build.  (GPU sanitizers are not available on the MI355X pool.)
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=/tmp/msmhip_asan
mkdir -p $OUT
cd $ROOT/newmsm_amd/csrc
make >/dev/null
for f in pool.cpp stager.cpp host_mesh.cpp octree.cpp api.cpp cost.cpp cost_cliques.cpp group.cpp regtools.cpp; do
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -x hip -O1 -g -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -fsanitize=address -fno-omit-frame-pointer -c $f -o $OUT/${f%.cpp}.o
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -fsanitize=address -shared-libsan -o $OUT/libmsmhip.so $OUT/*.o kernels.o octree_kernels.o resample_kernels.o unary_kernels.o clique_kernels.o move_kernels.o group_kernels.o
RT=$(ls /opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.asan-x86_64.so | head -1)
cd $ROOT
LD_PRELOAD=$RT ASAN_OPTIONS=detect_leaks=0:verify_asan_link_order=0 MSM_LIB_PATH=$OUT/libmsmhip.so python -m pytest tests/test_host_logic.py tests/test_abi.py tests/test_meshio.py tests/test_golden.py tests/test_anatomy_grid.py tests/test_config.py -x -q -m "not gpu"
