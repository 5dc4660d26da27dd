#!/bin/bash
# usage: tools/kernel_stats.sh TAG [LIBPATH] -- per-kernel average time (rocprofv3 --kernel-trace --stats) of the unary kernels in
# bench.py, optionally with another build of the library (MSM_LIB_PATH) for A/B comparisons of kernel variants
export TMPDIR=/tmp
export MSM_BENCH_NOCHECK=1
tag=$1
MSM_LIB_PATH=${2:-} timeout -k 5 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$tag -- python3 bench.py --steps 50 --warmup 5 --no-cpu-baseline > gpurun_out/prof_$tag.log 2>&1
f=$(find gpurun_out/prof_$tag -name "*kernel_stats.csv" | head -1)
grep "k_unary" $f | awk -F'",' '{split($2,a,","); print $1, a[3]}' | cut -c1-120
grep -o '"ms_per_step": [0-9.]*' gpurun_out/prof_$tag.log
