#!/bin/bash
# End-of-round evidence on the GPU box (one gpurun call): full GPU test suite, the bench line, rocprofv3 kernel stats of the
# same command, HBM traffic (two PMC passes) and the SQ / LDS counters of the conv kernels.  Everything lands under
# gpurun_out/$TAG/; tools/pmc_traffic.py and tools/pmc_sq.py turn the PMC CSVs into the summaries kept in profiles/.
#   tools/round_profiles.sh r02
set -o pipefail
TAG=${1:-rXX}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
rm -rf $OUT  # (a previous call's per-pid CSVs would be picked up by the summarising scripts)
mkdir -p $OUT
cd $ROOT
echo "== tests"; timeout -k 10 900 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1; rc=$?; tail -3 $OUT/pytest_gpu.log
[ $rc -ne 0 ] && { echo "GPU tests failed"; exit 1; }
echo "== bench"; timeout -k 10 600 python bench.py > $OUT/bench_1gpu.json 2> $OUT/bench_1gpu.err || exit 1
python tools/show_bench.py $OUT/bench_1gpu.json
cd /tmp && export TMPDIR=/tmp
# the profiled runs skip bench.py's device wake-up steps: the step counts in the summaries stay 80 (10 + 50 + 20) and 25 (5 + 20)
export FOSVOS_BENCH_PRECONDITION=0
echo "== kernel stats"
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $ROOT/bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-infer --no-variants --no-alone > $OUT/bench_under_rocprof.json 2> $OUT/stats.err || exit 1
echo "== pmc traffic"
timeout -k 10 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 $ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-infer --no-roofline --no-variants > $OUT/pmc_fetch.json 2> $OUT/pmc_fetch.err || exit 1
timeout -k 10 600 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 $ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-infer --no-roofline --no-variants > $OUT/pmc_write.json 2> $OUT/pmc_write.err || exit 1
echo "== pmc sq"
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS" "SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_INSTS_LDS" "GRBM_GUI_ACTIVE SQ_INST_LEVEL_VMEM SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_RD"; do
  i=$((i+1))
  BENCH_CONV_N=5 timeout -k 10 300 rocprofv3 --pmc $set --output-format csv -d $OUT/pmc_sq$i -- python3 $ROOT/tools/bench_conv.py 2 conv fwd,dgrad,wgrad > $OUT/pmc_sq$i.log 2>&1 || { echo "sq pass $i failed"; tail -3 $OUT/pmc_sq$i.log; }
done
cd $ROOT
python tools/pmc_traffic.py $OUT/pmc_fetch $OUT/pmc_write $OUT/pmc_traffic.json 25 && echo traffic ok
python tools/pmc_sq.py $(ls $OUT/pmc_sq*/*/*counter_collection.csv 2>/dev/null) > $OUT/pmc_sq.txt 2>&1; head -12 $OUT/pmc_sq.txt
# keep what is merged back small: the raw per-dispatch CSVs are large
find $OUT -name "*counter_collection.csv" -size +20M -delete
du -sh $OUT
