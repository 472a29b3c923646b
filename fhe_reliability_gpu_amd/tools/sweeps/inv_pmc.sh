# round 3: forward against inverse passes on the headline batch, kernel times and three counter sets (each in its own pass)
set -e
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/invp
i=0
for set in "" "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU" "SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_WAIT_INST_LDS" "SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_WAIT_ANY"; do
  i=$((i+1))
  rm -rf gpurun_out/invp/p
  if [ -z "$set" ]; then
    rocprofv3 --kernel-trace -d gpurun_out/invp/p -o r -- python3 fhe_reliability_gpu_amd/tools/inv_rate.py > gpurun_out/invp/run_$i.log 2>&1 || true
  else
    rocprofv3 --kernel-trace --pmc $set -d gpurun_out/invp/p -o r -- python3 fhe_reliability_gpu_amd/tools/inv_rate.py > gpurun_out/invp/run_$i.log 2>&1 || true
  fi
  python3 profiles/rocpd_summary.py $(find gpurun_out/invp/p -name "*.db" | head -1) 1000 > gpurun_out/invp/pass_$i.txt || true
  rm -rf gpurun_out/invp/p
done
ls gpurun_out/invp
