#!/bin/bash
# tools/collect_registration_cpp_profile.sh TAG -- on the GPU box: per-kernel times (rocprofv3 --kernel-trace --stats) of the HCP MSMAll-shaped
# registration driven by the compiled C++ host (tools/cpp/registration_bench: no Python in the traced process); writes gpurun_out/TAG_kernel_stats.csv
export TMPDIR=/tmp
tag=$1
python3 -c "import __graft_entry__ as g; g.build()" || exit 1
python3 tools/make_registration_inputs.py gpurun_out/reg_inputs HCP_MSMAll 32 || exit 1
timeout -k 5 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$tag -- tools/cpp/registration_bench gpurun_out/reg_inputs/in.bag gpurun_out/reg_inputs/out.bag gpurun_out/reg_inputs/conf 2 > gpurun_out/prof_$tag.log 2>&1 || exit 1
cp "$(find gpurun_out/prof_$tag -name '*kernel_stats.csv' | head -1)" gpurun_out/${tag}_kernel_stats.csv
tail -1 gpurun_out/prof_$tag.log | cut -c1-400
head -8 gpurun_out/${tag}_kernel_stats.csv | cut -c1-160
rm -rf gpurun_out/reg_inputs
