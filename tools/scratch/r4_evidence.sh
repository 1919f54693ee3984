#!/bin/bash
# round-4 evidence pass: run ON THE GPU BOX from the repo root; everything lands in gpurun_out/r4ev/
R=$PWD; O=$R/gpurun_out/r4ev; mkdir -p $O; V=${1:-v2}
export TMPDIR=/tmp
timeout 300 python bench.py > $O/r04_bench_l_default_$V.json.log 2>&1
timeout 200 python tools/conv_bench.py > $O/r04_conv_table_$V.txt 2>&1
(cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/prof_kt -o r -- python $R/bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-roofline > $O/kt_bench.log 2>&1)
DB=$(find /tmp/prof_kt -name "*results.db" | head -1)
python tools/rocpd_counts.py $DB 12 --by-time > $O/r04_bench_l_graph_kernel_stats_steady_$V.txt 2>&1
python tools/rocpd_counts.py $DB 12 > $O/r04_launch_counts_$V.txt 2>&1
timeout 500 python tools/pmc_traffic.py > $O/pmc_traffic.log 2>&1
timeout 500 python tools/pmc_kernels.py > $O/pmc_kernels.log 2>&1
cp gpurun_out/r04_pmc_*eager.txt $O/ 2>/dev/null; cp gpurun_out/r04_pmc_*eager.txt profiles/ 2>/dev/null
timeout 300 python tools/gemm_p8_bench.py 5 --quick > $O/r04_gemm_p8_table_$V.txt 2>&1
timeout 200 python tools/bench_backbone.py > $O/r04_backbone_7b_1024_v1.log 2>&1
timeout 200 python tools/bench_inference.py > $O/r04_inference_sliding_window_v1.json.log 2>&1
timeout 100 python tools/scratch/msda_time.py > $O/r04_msda_time_$V.txt 2>&1
DU_MSDA_NO_PLANE=1 timeout 100 python tools/scratch/msda_time.py >> $O/r04_msda_time_$V.txt 2>&1
timeout 100 python tools/scratch/memcpy_bw.py > $O/r04_memcpy_bw.txt 2>&1
timeout 300 python bench.py > $O/r04_bench_l_default_${V}b.json.log 2>&1      # second line, now with the PMC summaries of this very build in place
tail -c 600 $O/r04_bench_l_default_${V}b.json.log; echo; ls -la $O
