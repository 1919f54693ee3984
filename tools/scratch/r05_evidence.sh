export TMPDIR=/tmp; R=$PWD; O=$R/gpurun_out/e1; mkdir -p $O
timeout 420 python tools/ab_bench.py --rounds 3 --steps 30 --arm r04=build/r04 --arm r05=. --out $O/ab.txt > $O/ab.log 2>&1; echo ab rc $?; tail -4 $O/ab.txt
timeout 200 python bench.py > $O/bench_default.json.log 2> $O/bench_default.err; echo bench rc $?
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof -o p -- python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline > $O/trace_bench.log 2>&1); echo trace rc $?
DB=$(find /tmp/prof -name "*results.db" | head -1); python tools/rocpd_counts.py $DB 20 --by-time > $O/steady.txt 2>&1; python tools/rocpd_counts.py $DB 20 > $O/launch_counts.txt 2>&1; head -3 $O/steady.txt | cut -c1-200
timeout 400 python tools/pmc_traffic.py > $O/pmc_traffic.log 2>&1; echo pmc_traffic rc $?
timeout 400 python tools/pmc_kernels.py > $O/pmc_kernels.log 2>&1; echo pmc_kernels rc $?
cp gpurun_out/r05_pmc_* $O/ 2>/dev/null
timeout 200 python tools/bench_backbone.py > $O/backbone_7b.log 2>&1; echo backbone rc $?; tail -1 $O/backbone_7b.log | cut -c1-200
timeout 200 python tools/bench_inference.py > $O/inference.log 2>&1; echo inference rc $?; tail -1 $O/inference.log | cut -c1-300
