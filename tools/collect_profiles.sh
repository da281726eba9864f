#!/bin/bash
# Collect the rocprofv3 evidence of one round on the GPU box and reduce it into gpurun_out/<tag>/ (copy what is to be judged into
# profiles/ afterwards).  usage (from the repository root on the GPU box):  bash tools/collect_profiles.sh r03
# Counter passes run with --tune 200,1301 (eager launches, a host wait per layer): rocprofv3's counter collection on this image
# faults once a few hundred dispatches are queued un-waited (tools/pmc_probe.py).  No --pmc run is combined with any trace domain
# other than --kernel-trace.
set -u
TAG=${1:-r04}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
P=/tmp/prof_$TAG
rm -rf "$P"

echo "[1] kernel trace + stats of the default bench run"
rocprofv3 --kernel-trace --stats --output-format csv -d $P/a -- python3 $ROOT/bench.py --no-cpu-baseline > $OUT/a_bench.log 2>&1
python3 $ROOT/tools/summarize_trace.py $P/a $OUT/a_by_grid.csv > /dev/null
find $P/a -name "*kernel_stats.csv" -exec cp {} $OUT/a_kernel_stats_raw.csv \;
grep -v "at::native\|rocclr\|Cijk_\|elementwise\|vectorized" $OUT/a_kernel_stats_raw.csv > $OUT/a_kernel_stats.csv
tail -1 $OUT/a_bench.log | cut -c1-400

echo "[2] kernel trace of three un-instrumented steps (time per family)"
rocprofv3 --kernel-trace --output-format csv -d $P/t -- python3 $ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline > $OUT/t_bench.log 2>&1

echo "[3] FETCH_SIZE over one step"
timeout -k 10 400 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $P/f -- python3 $ROOT/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-roofline --tune 200,1301 > $OUT/f_bench.log 2>&1
F=$(find $P/f -name "*counter_collection.csv" | head -1)
python3 $ROOT/tools/pmc_step_split.py "$F" 1.7b $OUT/pmc_fetch_bench_step.json > /dev/null && echo "  fetch json ok"

echo "[4] MFMA busy over one step"
timeout -k 10 400 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $P/m -- python3 $ROOT/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-roofline --tune 200,1301 > $OUT/m_bench.log 2>&1
python3 $ROOT/tools/pmc_mfma_summary.py $P/m $OUT/pmc_mfma_busy.csv > /dev/null && echo "  mfma csv ok"

echo "[5] families"
python3 $ROOT/tools/family_summary.py --trace $P/t --trace-steps 3 --fetch $P/f --mfma $P/m --json $OUT/pmc_fetch_bench_step.json | tee $OUT/families.txt

echo "[6] L1 / L2 read requests of the decode GEMM shapes (A operand against weights)"
timeout -k 10 300 rocprofv3 --kernel-trace --pmc TCP_TCC_READ_REQ_sum TCP_TOTAL_READ_sum TCC_HIT_sum TCC_MISS_sum --output-format csv -d $P/l -- python3 $ROOT/tools/bench_gemm_col.py > $OUT/l_bench.log 2>&1
python3 $ROOT/tools/pmc_by_kernel.py $P/l k_gemm_col > $OUT/pmc_gemm_col_l1.csv 2>/dev/null && head -20 $OUT/pmc_gemm_col_l1.csv
echo "[7] L1 / L2 read requests of the decode attention (one bench step)"
timeout -k 10 400 rocprofv3 --kernel-trace --pmc TCP_TCC_READ_REQ_sum TCP_TOTAL_READ_sum TCC_HIT_sum TCC_MISS_sum --output-format csv -d $P/q -- python3 $ROOT/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-roofline --tune 200,1301 > $OUT/q_bench.log 2>&1
python3 $ROOT/tools/pmc_by_kernel.py $P/q k_attention > $OUT/pmc_attention.csv 2>/dev/null && cat $OUT/pmc_attention.csv
echo done
