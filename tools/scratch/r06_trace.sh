# steady-state kernel table of the replayed step + per-shape eager table + glue call sites (round 6 working runs)
export TMPDIR=/tmp; R=$PWD; O=$R/gpurun_out/${1:-t1}; mkdir -p $O
(cd /tmp && rm -rf /tmp/prof && timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof -o p -- python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline > $O/trace_bench.log 2>&1); echo trace rc $?
DB=$(find /tmp/prof -name "*results.db" | head -1); python tools/rocpd_counts.py $DB 20 --by-time > $O/steady.txt 2>&1; python tools/rocpd_counts.py $DB 20 > $O/launch_counts.txt 2>&1; head -3 $O/steady.txt | cut -c1-200
timeout 200 python tools/step_detail.py > $O/step_detail.txt 2>&1; echo step_detail rc $?
timeout 200 python tools/glue_trace.py --trainstep --top 80 > $O/glue.txt 2>&1; echo glue rc $?
