#!/bin/bash
# tools/collect_query_profile.sh TAG -- on the GPU box: per-kernel times of the search kernels (rocprofv3 --kernel-trace --stats) for
# tools/time_query.py and tools/time_resample.py; writes gpurun_out/TAG_query_kernel_stats.csv / gpurun_out/TAG_resample_kernel_stats.csv
export TMPDIR=/tmp
tag=$1
timeout -k 5 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_${tag}_query -- python3 tools/time_query.py 20 > gpurun_out/${tag}_query.log 2>&1 || exit 1
cp "$(find gpurun_out/prof_${tag}_query -name '*kernel_stats.csv' | head -1)" gpurun_out/${tag}_query_kernel_stats.csv
timeout -k 5 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_${tag}_resample -- python3 tools/time_resample.py > gpurun_out/${tag}_resample.log 2>&1 || exit 1
cp "$(find gpurun_out/prof_${tag}_resample -name '*kernel_stats.csv' | head -1)" gpurun_out/${tag}_resample_kernel_stats.csv
