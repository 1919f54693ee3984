#!/bin/bash
# usage: attn_pmc_run.sh "impl var N" ...   -> per configuration, two PMC passes, summed over the attention kernel's launches
cd /tmp; export TMPDIR=/tmp
PA="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE"
PB="SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INST_CYCLES_VMEM SQ_WAVES"
for cfg in "$@"; do
  tag=$(echo $cfg | tr ' ' '_')
  for p in A B; do
    if [ $p = A ]; then C="$PA"; else C="$PB"; fi
    rm -rf /tmp/pmc_$tag$p
    rocprofv3 --pmc $C --kernel-trace --output-format csv -d /tmp/pmc_$tag$p -o p -- python $GRAFT_REPO_ROOT/tools/scratch/attn_pmc2.py $cfg > /tmp/pmc_$tag$p.log 2>&1 || tail -5 /tmp/pmc_$tag$p.log
  done
  python - "$tag" <<'PY'
import csv, glob, sys, collections
tag = sys.argv[1]
tot = collections.defaultdict(float); n = collections.Counter()
for p in "AB":
    for f in glob.glob(f"/tmp/pmc_{tag}{p}/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            if "attn_fwd" not in row["Kernel_Name"]: continue
            tot[row["Counter_Name"]] += float(row["Counter_Value"]); n[row["Counter_Name"]] += 1
if not tot: print(tag, "no data"); sys.exit()
L = max(n.values())
g = lambda k: tot.get(k, float("nan")) / max(n.get(k, 1), 1)
launches = 4
print(f"== {tag}: rows per counter {dict(n)}")
wc = tot["SQ_WAVE_CYCLES"]; 
print(f"   wave view: parked {tot['SQ_WAIT_ANY']/wc:.3f} stalled {tot['SQ_WAIT_INST_ANY']/wc:.3f} (lds {tot['SQ_WAIT_INST_LDS']/wc:.3f}) issuing {tot['SQ_ACTIVE_INST_ANY']/wc:.3f}")
gui = tot["GRBM_GUI_ACTIVE"] / n["GRBM_GUI_ACTIVE"]
print(f"   GUI_ACTIVE per launch {gui:.0f}; MFMA busy cycles per launch / (GUI x 1024 SIMDs): {tot['SQ_VALU_MFMA_BUSY_CYCLES']/launches/(gui*1024):.3f}; busy_cycles {tot['SQ_BUSY_CYCLES']/launches:.0f}")
for k in ("SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS", "SQ_ACTIVE_INST_VMEM", "SQ_ACTIVE_INST_SCA", "SQ_INST_CYCLES_VMEM"):
    print(f"   {k} x4 / (GUI x 1024): {4*tot[k]/launches/(gui*1024):.3f}")
print(f"   INSTS_VALU per launch {tot['SQ_INSTS_VALU']/launches:.0f}  INSTS_MFMA {tot.get('SQ_INSTS_MFMA',0)/launches:.0f}  waves {tot['SQ_WAVES']/launches:.0f}  wave_cycles {wc/launches:.0f}")
PY
done
