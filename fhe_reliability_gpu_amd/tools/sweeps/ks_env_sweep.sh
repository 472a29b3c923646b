# round 3: kernel times inside a key switch for several settings of one environment knob
#   usage: ks_env_sweep.sh VAR "v1 v2 ..." "grep pattern" [ks_loop args]
set -e
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
VAR=$1; VALS=$2; PAT=$3; shift 3
ARGS=${@:-16 44 11 4 10}
for v in $VALS; do
  export $VAR=$v
  rm -rf gpurun_out/sw_$v
  rocprofv3 --kernel-trace -d gpurun_out/sw_$v -o r -- python3 fhe_reliability_gpu_amd/tools/ks_loop.py $ARGS > gpurun_out/sw_$VAR_$v.log 2>&1
  python3 profiles/rocpd_summary.py $(find gpurun_out/sw_$v -name "*.db" | head -1) > gpurun_out/sw_${VAR}_$v.txt
  echo "== $VAR=$v ($ARGS)"; grep -i "$PAT" gpurun_out/sw_${VAR}_$v.txt | cut -c1-150
  rm -rf gpurun_out/sw_$v
done
