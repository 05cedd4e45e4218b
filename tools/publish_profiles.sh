#!/bin/bash
# Copy the files of a `tools/collect_profiles.sh <tag> r04` + `<tag> r04base` run from gpurun_out/<tag>/ into profiles/r04_* (the tracked
# evidence set) and rebuild profiles/pmc_traffic.json.      usage: tools/publish_profiles.sh <tag>
set -e
O=gpurun_out/$1
python3 tools/pmc_summarize.py --fetch $O/pmc_f/f_counter_collection.csv --write $O/pmc_w/w_counter_collection.csv --skip-first 8 --tag r04_tiny --out profiles | tail -1
cp $O/tiny_bench.json profiles/r04_tiny_bench.json
cp $O/tiny_bench_inflight1.json profiles/r04_tiny_bench_inflight1.json
cp $O/tiny_bench_under_rocprof.json profiles/r04_tiny_bench_under_rocprof.json
cp $O/tiny_bench_inflight1_under_rocprof.json profiles/r04_tiny_bench_inflight1_under_rocprof.json
cp $O/prof_if2/p_kernel_stats.csv profiles/r04_tiny_kernel_stats.csv
cp $O/prof_if1/p_kernel_stats.csv profiles/r04_tiny_inflight1_kernel_stats.csv
cp $O/prof_s3/p_kernel_stats.csv profiles/r04_exact_index_split3_kernel_stats.csv
cat $O/exact_index_split3.txt $O/exact_index_fp32.txt | grep value > profiles/r04_exact_index.txt
cp $O/base_bench.json profiles/r04_base_bench.json
cp $O/base5_bench.json profiles/r04_base5_bench.json
cp $O/base5_decoder_only_bench.json profiles/r04_base5_decoder_only_bench.json
cp $O/prof_base/p_kernel_stats.csv profiles/r04_base_inflight1_kernel_stats.csv
cp $O/prof_base5/p_kernel_stats.csv profiles/r04_base5_kernel_stats.csv
cp $O/base_bench_inflight1_under_rocprof.json profiles/r04_base_bench_inflight1_under_rocprof.json
cp $O/base5_bench_inflight1_under_rocprof.json profiles/r04_base5_bench_inflight1_under_rocprof.json
