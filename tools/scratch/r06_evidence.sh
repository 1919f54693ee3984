# round-6 evidence on one box: interleaved A/B against the round-5 tree, the default bench line, the steady-state kernel table of the
# replayed step (+ launch counts, per-shape eager table), PMC traffic / cycle passes on the shipped digest, config 5, sliding-window inference
export TMPDIR=/tmp; R=$PWD; O=$R/gpurun_out/${1:-e1}; mkdir -p $O
timeout 500 python tools/ab_bench.py --rounds 4 --steps 30 --arm r05=build/r05 --arm r06=. --out $O/ab.txt > $O/ab.log 2>&1; echo ab rc $?; tail -3 $O/ab.txt
timeout 300 python bench.py > $O/bench_default.json.log 2> $O/bench_default.err; echo bench rc $?
(cd /tmp && rm -rf /tmp/prof && timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof -o p -- python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline > $O/trace_bench.log 2>&1); echo trace rc $?
DB=$(find /tmp/prof -name "*results.db" | head -1); python tools/rocpd_counts.py $DB 20 --by-time > $O/steady.txt 2>&1; python tools/rocpd_counts.py $DB 20 > $O/launch_counts.txt 2>&1; head -3 $O/steady.txt | cut -c1-200
timeout 200 python tools/step_detail.py > $O/step_detail.txt 2>&1; echo step_detail rc $?
timeout 500 python tools/pmc_traffic.py > $O/pmc_traffic.log 2>&1; echo pmc_traffic rc $?
timeout 500 python tools/pmc_kernels.py > $O/pmc_kernels.log 2>&1; echo pmc_kernels rc $?
cp gpurun_out/r06_pmc_* $O/ 2>/dev/null; ls gpurun_out | grep pmc | head
timeout 200 python tools/bench_backbone.py > $O/backbone_7b.log 2>&1; echo backbone rc $?; tail -1 $O/backbone_7b.log | cut -c1-200
timeout 200 python tools/bench_inference.py > $O/inference.log 2>&1; echo inference rc $?; tail -1 $O/inference.log | cut -c1-300
timeout 300 python tools/gemm_p8_bench.py 7 --quick > $O/gemm_p8_table.txt 2>&1; echo gemm table rc $?
