# round 3: counters of the single-launch transform (csrc/ntt_fused.hip, --mode fused) on the headline batch, each PMC set in
# a pass of its own with --kernel-trace only.  Run on the GPU box from the repo root:  bash profiles/r03_fused_pmc.sh
set -e
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r03f
mkdir -p $OUT
sum() { python3 profiles/rocpd_summary.py $(find $1 -name "*.db" | head -1) ${2:-0}; }
CMD="python3 bench.py --mode fused --no-cpu --no-extras --steps 5 --warmup 2 --chunk-mib 0"
$CMD > $OUT/fused_bench.json 2> $OUT/fused_bench.err || true
i=0
for set in "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum" "FETCH_SIZE WRITE_SIZE" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $set -d $OUT/p -o r -- $CMD > /dev/null 2>&1 || true
  sum $OUT/p 100000 > $OUT/fused_pmc_$i.txt || true
  rm -rf $OUT/p
done
# the two-launch default on the same batch as one launch pair per step (no sub-batching), same counters, for the comparison
CMD2="python3 bench.py --no-cpu --no-extras --steps 5 --warmup 2 --chunk-mib 0"
i=0
for set in "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $set -d $OUT/p -o r -- $CMD2 > /dev/null 2>&1 || true
  sum $OUT/p 100000 > $OUT/twopass_pmc_$i.txt || true
  rm -rf $OUT/p
done
ls -la $OUT
