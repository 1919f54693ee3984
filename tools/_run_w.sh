set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
( timeout 420 python -m pytest tests -m gpu -x -q > gpurun_out/r2_gputest_w.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r2_gputest_w.log )
grep -E "passed|failed|error" gpurun_out/r2_gputest_w.log | tail -3
( timeout 240 python tools/pmc_kernels.py > gpurun_out/r2_pmc_sq_w.log 2>&1; echo "rc=$?" >> gpurun_out/r2_pmc_sq_w.log )
tail -5 gpurun_out/r2_pmc_sq_w.log
timeout 200 python bench.py > gpurun_out/r2_bench_default_w.log 2>&1; tail -1 gpurun_out/r2_bench_default_w.log | cut -c1-400
timeout 120 python tools/bench_backbone.py > gpurun_out/r2_backbone_7b_w.log 2>&1; tail -1 gpurun_out/r2_backbone_7b_w.log | cut -c1-600
timeout 120 python tools/glue_trace.py --trainstep --top 120 > gpurun_out/r2_glue_ts_w.txt 2>&1; head -3 gpurun_out/r2_glue_ts_w.txt
