# per-kernel steady tables of the replayed step for two configurations (same box, back to back)
export TMPDIR=/tmp; R=$PWD; O=$R/gpurun_out/${1:-tab}; mkdir -p $O
for cfg in base heads; do
  if [ $cfg = heads ]; then export DINOUNET_AB_KNOBS=1 DINOUNET_QKV_HEADS=1; fi
  (cd /tmp && rm -rf /tmp/prof_$cfg && timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_$cfg -o p -- python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline > $O/trace_$cfg.log 2>&1); echo trace $cfg rc $?
  DB=$(find /tmp/prof_$cfg -name "*results.db" | head -1); python tools/rocpd_counts.py $DB 20 --by-time > $O/steady_$cfg.txt 2>&1
  grep -E "kernel time|pp_kernel|attn_fwd|rope|p8n_kernel<float" $O/steady_$cfg.txt | cut -c1-150
done
