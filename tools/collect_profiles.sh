#!/bin/bash
# Run ON THE GPU BOX (through gpurun) from the repository root: the round's measurement artefacts into gpurun_out/<tag>_*.
#   bash tools/collect_profiles.sh r02
# Kernel-trace statistics and the two PMC passes are separate rocprofv3 runs (PMC passes carry --kernel-trace only).
set -o pipefail
TAG=${1:-rXX}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
python3 $ROOT/bench.py > $OUT/${TAG}_bench_full.json 2> $OUT/${TAG}_bench_full.err || exit 1
echo "[collect] bench done"
B="$ROOT/bench.py --no-cpu-baseline --train-steps 0 --no-vae --no-full-call --no-phosc"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_prof_bench -o b -- python3 $B > $OUT/${TAG}_prof_bench.log 2>&1 || exit 2
echo "[collect] bench kernel stats done"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_prof_train -o t -- python3 $ROOT/tools/train_prof.py > $OUT/${TAG}_prof_train.log 2>&1 || exit 3
echo "[collect] train kernel stats done"
P="$ROOT/bench.py --steps 10 --warmup 2 --no-cpu-baseline --train-steps 0 --no-vae --no-roofline --no-full-call --no-phosc"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/${TAG}_pmc_fetch -o f -- python3 $P > $OUT/${TAG}_pmc_fetch.log 2>&1 || exit 4
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/${TAG}_pmc_write -o w -- python3 $P > $OUT/${TAG}_pmc_write.log 2>&1 || exit 5
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/${TAG}_pmc_mfma -o m -- python3 $P > $OUT/${TAG}_pmc_mfma.log 2>&1 || exit 7
M=$(find $OUT/${TAG}_pmc_mfma -name "*counter_collection.csv" | head -1)
python3 $ROOT/tools/pmc_mfma.py "$M" > $OUT/${TAG}_pmc_mfma_util.json || exit 8
F=$(find $OUT/${TAG}_pmc_fetch -name "*counter_collection.csv" | head -1)
W=$(find $OUT/${TAG}_pmc_write -name "*counter_collection.csv" | head -1)
python3 $ROOT/tools/pmc_traffic.py "$F" "$W" > $OUT/${TAG}_pmc_hbm_traffic.json || exit 6
for d in bench train; do
    f=$(find $OUT/${TAG}_prof_$d -name "*kernel_stats.csv" | head -1)
    [ -n "$f" ] && cp "$f" $OUT/${TAG}_${d}_kernel_stats.csv
done
# the raw traces are large: keep the summaries only
find $OUT/${TAG}_prof_bench $OUT/${TAG}_prof_train $OUT/${TAG}_pmc_fetch $OUT/${TAG}_pmc_write -name "*kernel_trace.csv" -delete
find $OUT/${TAG}_pmc_fetch $OUT/${TAG}_pmc_write $OUT/${TAG}_pmc_mfma -name "*counter_collection.csv" -delete
find $OUT/${TAG}_pmc_mfma -name "*kernel_trace.csv" -delete
# the per-launch tables and the in-graph timeline of one step
python3 $ROOT/tools/step_ops.py > $OUT/${TAG}_step_ops.txt 2>/dev/null
VARIANT=phosc python3 $ROOT/tools/step_ops.py > $OUT/${TAG}_step_ops_phosc.txt 2>/dev/null
rocprofv3 --kernel-trace --output-format csv -d $OUT/${TAG}_tl -o t -- python3 $ROOT/bench.py --steps 40 --warmup 5 --no-cpu-baseline --train-steps 0 --no-vae --no-roofline --no-full-call --no-phosc > $OUT/${TAG}_tl.log 2>&1
python3 $ROOT/tools/step_timeline.py $(find $OUT/${TAG}_tl -name "*kernel_trace.csv" | head -1) > $OUT/${TAG}_step_timeline.txt
rm -rf $OUT/${TAG}_tl
# the training step: per-launch table, the weight-gradient kernel alone, MFMA-busy counters of its kernels
python3 $ROOT/tools/train_ops.py > $OUT/${TAG}_train_ops.txt 2>/dev/null
python3 $ROOT/tools/dw_bench.py > $OUT/${TAG}_dw_bench.txt 2>/dev/null
python3 $ROOT/tools/dw_bench.py --shape lin8x32,skip8x32 --group 6 >> $OUT/${TAG}_dw_bench.txt 2>/dev/null
STEPS=6 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/${TAG}_pmc_mfma_train -o m -- python3 $ROOT/tools/train_prof.py > $OUT/${TAG}_pmc_mfma_train.log 2>&1
M=$(find $OUT/${TAG}_pmc_mfma_train -name "*counter_collection.csv" | head -1)
python3 $ROOT/tools/pmc_mfma.py "$M" > $OUT/${TAG}_pmc_mfma_util_train.json
rm -rf $OUT/${TAG}_pmc_mfma_train
# shader clock / socket power under the fused feed-forward on all CUs and on eight workgroups (the power cap)
{ echo "wd_ff_fused, m = 16384 (256 workgroups): count, sclk, socket W"; bash $ROOT/tools/clock_watch.sh 15 python3 $ROOT/tools/ff_bench.py --proj 1 --m 16384 --iters 250000;
  echo "wd_ff_fused, m = 512 (8 workgroups)"; bash $ROOT/tools/clock_watch.sh 15 python3 $ROOT/tools/ff_bench.py --proj 1 --m 512 --iters 350000; } > $OUT/${TAG}_clock_power.txt 2>/dev/null
echo "[collect] done"
ls $OUT | grep "^${TAG}_"
