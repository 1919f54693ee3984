export TMPDIR=/tmp; R=$PWD; O=$R/gpurun_out/e2; mkdir -p $O
timeout 500 python tools/ab_bench.py --rounds 3 --steps 30 --arm r04=build/r04 --arm r05=. --arm r05_chains2=.,DINOUNET_VIT_CHAINS=2 --out $O/ab.txt > $O/ab.log 2>&1; echo ab rc $?; tail -4 $O/ab.txt
timeout 200 python bench.py > $O/bench_default.json.log 2> $O/bench_default.err; echo bench rc $?
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof4 -o p -- python $R/build/r04/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline > $O/trace_r04.log 2>&1); echo trace r04 rc $?
DB=$(find /tmp/prof4 -name "*results.db" | head -1); python tools/rocpd_counts.py $DB 20 --by-time > $O/steady_r04.txt 2>&1
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof5 -o p -- python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline > $O/trace_r05.log 2>&1); echo trace r05 rc $?
DB=$(find /tmp/prof5 -name "*results.db" | head -1); python tools/rocpd_counts.py $DB 20 --by-time > $O/steady_r05.txt 2>&1
head -2 $O/steady_r04.txt | cut -c1-120; head -2 $O/steady_r05.txt | cut -c1-120
find /tmp/prof5 -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/kernel_stats_r05.csv
