#!/bin/bash
# tools/collect_query_pmc.sh TAG -- on the GPU box: the PMC passes of MI355X_MICROARCH.md (one counter set per run, never together with API traces) for the
# search kernels of tools/time_query.py; writes gpurun_out/TAG_kernel_stats.csv and gpurun_out/TAG_pmc.json (tools/summarise_pmc.py)
export TMPDIR=/tmp
tag=$1
B="python3 tools/time_query.py 20"
run() { timeout -k 5 150 rocprofv3 "$@" -- $B > /dev/null 2>&1; }
timeout -k 5 150 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$tag -- $B > gpurun_out/prof_$tag.log 2>&1 || exit 1
run --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_VMEM_RD --output-format csv -d gpurun_out/pmc_${tag}_1 || exit 1
run --kernel-trace --pmc SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_THREAD_CYCLES_VALU SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_WR --output-format csv -d gpurun_out/pmc_${tag}_2 || exit 1
run --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_${tag}_3 || exit 1
run --kernel-trace --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d gpurun_out/pmc_${tag}_4 || exit 1
run --kernel-trace --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/pmc_${tag}_5 || exit 1
cp "$(find gpurun_out/prof_$tag -name '*kernel_stats.csv' | head -1)" gpurun_out/${tag}_kernel_stats.csv
python3 tools/summarise_pmc.py $tag msm::k_query "$B"
