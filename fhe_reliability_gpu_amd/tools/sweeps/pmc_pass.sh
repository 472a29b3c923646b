# round 3: one or more PMC passes over a key-switch loop, kernel filter by name
#   usage: pmc_pass.sh "<grep pattern>" "<counters pass 1>" ["<counters pass 2>" ...]   (KS_ARGS overrides the shape)
set -e
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
PAT=$1; shift
i=0
for set in "$@"; do
  i=$((i+1))
  rm -rf gpurun_out/pmcp_$i
  rocprofv3 --kernel-trace --pmc $set -d gpurun_out/pmcp_$i -o r -- python3 fhe_reliability_gpu_amd/tools/ks_loop.py ${KS_ARGS:-16 44 11 4 6} > gpurun_out/pmcp_$i.log 2>&1 || { tail -5 gpurun_out/pmcp_$i.log; }
  python3 profiles/rocpd_summary.py $(find gpurun_out/pmcp_$i -name "*.db" | head -1) 200 > gpurun_out/pmcp_$i.txt || true
  rm -rf gpurun_out/pmcp_$i
  grep -A1 "$PAT" gpurun_out/pmcp_$i.txt | cut -c1-260
done
