# round 3: parity of the fixed-size base-conversion kernels + per-variant kernel times inside a config-5 key switch
set -e
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python -m pytest tests/test_gpu_keyswitch.py tests/test_gpu_parity.py -x -q -m gpu -k "baseconv or keyswitch or rotate or base_conversion" > gpurun_out/bc_tests.log 2>&1 || { tail -30 gpurun_out/bc_tests.log; exit 1; }
tail -3 gpurun_out/bc_tests.log
for v in 0 12 14 22 24; do
  export FHE_BC_VARIANT=$v
  rm -rf gpurun_out/bc_$v
  rocprofv3 --kernel-trace -d gpurun_out/bc_$v -o r -- python3 fhe_reliability_gpu_amd/tools/ks_loop.py 16 44 11 4 10 > gpurun_out/bc_$v.log 2>&1
  python3 profiles/rocpd_summary.py $(find gpurun_out/bc_$v -name "*.db" | head -1) > gpurun_out/bc_$v.txt
  echo "== variant $v"; grep -i "baseconv\|bc_exact" gpurun_out/bc_$v.txt | cut -c1-160
  rm -rf gpurun_out/bc_$v
done
