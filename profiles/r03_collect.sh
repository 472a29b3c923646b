# round 3: the profiles committed under profiles/r03_* -- run on the GPU box from the repo root:  bash profiles/r03_collect.sh
# (kernel trace and every PMC set in a pass of its own, --kernel-trace only, as the pool requires; summaries by profiles/rocpd_summary.py)
set -e
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r03
mkdir -p $OUT
sum() { python3 profiles/rocpd_summary.py $(find $1 -name "*.db" | head -1) ${2:-0}; }
DRIVER="python3 bench.py --gpus 1 --steps 20 --warmup 5"
# 1. the driver's exact command: the JSON line, then its kernel trace (headline loop rows: 3072- / 2048-workgroup sub-batch launches)
$DRIVER > $OUT/r03_bench.json 2> $OUT/r03_bench.err
rocprofv3 --kernel-trace --stats -d $OUT/kt -o r -- $DRIVER --no-cpu > /dev/null 2>&1
sum $OUT/kt 100000 > $OUT/r03_bench_kt.txt
rm -rf $OUT/kt
# 2. PMC passes on the headline loop only (no extras: the counters are per dispatch, the loop's kernels are the same)
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c -d $OUT/p -o r -- python3 bench.py --no-cpu --no-extras --steps 5 --warmup 2 > /dev/null 2>&1
  sum $OUT/p 100000 > $OUT/r03_bench_$(echo $c | tr A-Z a-z | sed s/_size//).txt
  rm -rf $OUT/p
done
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY -d $OUT/p -o r -- python3 bench.py --no-cpu --no-extras --steps 5 --warmup 2 > /dev/null 2>&1
sum $OUT/p 100000 > $OUT/r03_bench_sq.txt
rm -rf $OUT/p
# 3. the composites, kernel by kernel
for shape in "16 44 11 4" "16 44 4 11" "17 32 8 4" "16 16 4 4" "14 4 1 4"; do
  name=$(echo $shape | tr ' ' '_')
  rocprofv3 --kernel-trace -d $OUT/k -o r -- python3 fhe_reliability_gpu_amd/tools/ks_loop.py $shape 20 > /dev/null 2>&1
  sum $OUT/k > $OUT/r03_keyswitch_$name.kernels.txt
  rm -rf $OUT/k
done
rocprofv3 --kernel-trace -d $OUT/k -o r -- python3 fhe_reliability_gpu_amd/tools/ks_loop.py 16 44 11 4 20 rotate > /dev/null 2>&1
sum $OUT/k > $OUT/r03_rotate_16_44_11_4.kernels.txt
rm -rf $OUT/k
rocprofv3 --kernel-trace -d $OUT/k -o r -- python3 fhe_reliability_gpu_amd/tools/ks_loop.py 17 32 8 4 20 hmult > /dev/null 2>&1
sum $OUT/k > $OUT/r03_hmult_17_32_8_4.kernels.txt
rm -rf $OUT/k
rocprofv3 --kernel-trace -d $OUT/k -o r -- python3 fhe_reliability_gpu_amd/tools/ks_loop.py 16 44 11 4 5 hoisted > /dev/null 2>&1
sum $OUT/k > $OUT/r03_hoisted8_16_44_11_4.kernels.txt
rm -rf $OUT/k
# 4. counters of the key switch's two heaviest kernels
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_INSTS_VALU GRBM_GUI_ACTIVE -d $OUT/p -o r -- python3 fhe_reliability_gpu_amd/tools/ks_loop.py 16 44 11 4 6 > /dev/null 2>&1
sum $OUT/p 200 > $OUT/r03_keyswitch_16_44_11_4.sq.txt
rm -rf $OUT/p
ls -la $OUT
