# total device time per key switch for settings of one environment knob:  ks_total.sh VAR "v1 v2" shape...
set -e
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
VAR=$1; VALS=$2; shift 2
for v in $VALS; do
  export $VAR=$v
  rm -rf gpurun_out/kt_$v
  rocprofv3 --kernel-trace -d gpurun_out/kt_$v -o r -- python3 fhe_reliability_gpu_amd/tools/ks_loop.py "$@" > /dev/null 2>&1
  python3 profiles/rocpd_summary.py $(find gpurun_out/kt_$v -name "*.db" | head -1) | grep -v copyBuffer | awk -v tag="$VAR=$v ($*)" 'NR>1 {s += $1*$2} END {printf "%s: %.1f us per call (kernels only, /20 calls)\n", tag, s/20}'
  rm -rf gpurun_out/kt_$v
done
