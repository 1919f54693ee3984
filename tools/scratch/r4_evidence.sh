#!/bin/bash
# round-4 evidence pass: run ON THE GPU BOX from the repo root; everything lands in gpurun_out/r4ev/
R=$PWD; O=$R/gpurun_out/r4ev; mkdir -p $O; V=${1:-v1}
export TMPDIR=/tmp
timeout 300 python bench.py > $O/r04_bench_l_default_$V.json.log 2>&1
timeout 200 python tools/conv_bench.py > $O/r04_conv_table_$V.txt 2>&1
DU_CONV_STRIP=0 timeout 200 python tools/conv_bench.py > $O/r04_conv_table_halo_only_$V.txt 2>&1
(cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/prof_kt -o r -- python $R/bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-roofline > $O/kt_bench.log 2>&1)
DB=$(find /tmp/prof_kt -name "*results.db" | head -1)
python tools/rocpd_counts.py $DB 12 --by-time > $O/r04_bench_l_graph_kernel_stats_steady_$V.txt 2>&1
python tools/rocpd_counts.py $DB 12 > $O/r04_launch_counts_$V.txt 2>&1
find /tmp/prof_kt -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/r04_bench_l_graph_kernel_stats_$V.csv
timeout 500 python tools/pmc_traffic.py > $O/pmc_traffic.log 2>&1
timeout 500 python tools/pmc_kernels.py > $O/pmc_kernels.log 2>&1
cp gpurun_out/r04_pmc_*eager.txt $O/ 2>/dev/null
tail -c 1500 $O/r04_bench_l_default_$V.json.log
