#!/bin/bash
# rocprofv3 passes over bench.py (run on the GPU box): kernel stats, MFMA busy + clock, HBM fetch / write bytes.
#   usage: tools/profile_bench.sh <tag> [bench.py args...]     -> gpurun_out/prof_<tag>/{stats,mfma,fetch,write} + summaries
# Counters are collected in their own passes (no trace domains beside --kernel-trace); the program itself follows `--`.
set -e
tag=$1; shift
root=$(pwd)
out=$root/gpurun_out/prof_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
args="--steps 2 --warmup 1 --no-cpu-baseline $@"
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -o s -- python3 $root/bench.py $args > $out/stats.log 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES --kernel-trace --output-format csv -d $out/mfma -o m -- python3 $root/bench.py $args > $out/mfma.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $out/fetch -o f -- python3 $root/bench.py $args > $out/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $out/write -o w -- python3 $root/bench.py $args > $out/write.log 2>&1
cd $root
python3 tools/pmc_mfma.py $out/mfma $out/mfma_busy.json > $out/mfma_busy.txt
python3 tools/pmc_traffic.py $out/fetch $out/write $out/traffic.json > $out/traffic.txt
cp $(find $out/stats -name "*kernel_stats.csv" | head -1) $out/kernel_stats.csv
head -12 $out/kernel_stats.csv | cut -c1-150
head -8 $out/mfma_busy.txt
head -8 $out/traffic.txt
# keep the merge-back small: drop the raw per-dispatch CSVs
rm -rf $out/stats $out/mfma $out/fetch $out/write
