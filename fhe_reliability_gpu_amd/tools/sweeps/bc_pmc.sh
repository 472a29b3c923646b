# round 3: SQ counters of the key switch's kernels at config 5 (separate pmc passes, kernel trace only)
set -e
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_ACTIVE_INST_LDS" "GRBM_GUI_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_WAVES" "SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
  i=$((i+1))
  rm -rf gpurun_out/pmc_$i
  rocprofv3 --kernel-trace --pmc $set -d gpurun_out/pmc_$i -o r -- python3 fhe_reliability_gpu_amd/tools/ks_loop.py ${KS_ARGS:-16 44 11 4 6} > gpurun_out/pmc_$i.log 2>&1 || { tail -5 gpurun_out/pmc_$i.log; }
  python3 profiles/rocpd_summary.py $(find gpurun_out/pmc_$i -name "*.db" | head -1) 200 > gpurun_out/pmc_$i.txt || true
  rm -rf gpurun_out/pmc_$i
  grep -A1 "bc_exact\|rowmac" gpurun_out/pmc_$i.txt | cut -c1-230
done
