#!/bin/bash
# Round evidence run (GPU box):  bash tools/collect_profiles.sh <tag> <part>   -> gpurun_out/<tag>/...   (copy what is cited into profiles/)
# part tiny | counters | base | train.  rocprofv3: program right after `--`, counters in passes of their own (no --stats with --pmc).
set -e
TAG=${1:-r03}; PART=${2:-tiny}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/$TAG; mkdir -p $O
export TMPDIR=/tmp
cd /tmp
step() { echo "[collect $PART] $*"; }
if [ $PART = tiny2 ]; then
  step rocprof two chains; rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_if2 -o p -- python3 $R/bench.py --no-cpu-baseline --no-fp32-leg > $O/tiny_bench_under_rocprof.json 2> $O/prof_if2.log
  step rocprof one chain; rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_if1 -o p -- python3 $R/bench.py --in-flight 1 --no-cpu-baseline --no-fp32-leg > $O/tiny_bench_inflight1_under_rocprof.json 2> $O/prof_if1.log
  step stamps; TTV_LIB_PATH=$R/titok_video_amd/csrc/build/libtitok_hip_stamps.so python3 $R/tools/attn_stamps.py > $O/attn_stamps.txt 2>&1
  TTV_LIB_PATH=$R/titok_video_amd/csrc/build/libtitok_hip_stamps64.so python3 $R/tools/attn64_stamps.py > $O/attn64_stamps.txt 2>&1
  step attention alone; FP32=0 python3 $R/tools/attn_bench.py 1.5 6 > $O/attn_bench.txt 2>&1
  B=4 CLIP=32,256,256 K=1024 HQ=12 HKV=4 FP32=0 python3 $R/tools/attn_bench.py 1.5 >> $O/attn_bench.txt 2>&1
  step knock-outs of the 64-row kernel; for n in dma lds max exp all; do echo "== knock-out $n" >> $O/attn64_knockouts.txt; TTV_LIB_PATH=$R/titok_video_amd/csrc/build/libtitok_hip_ko64_$n.so FP32=0 python3 $R/tools/attn_bench.py 1.5 2>&1 | grep "w64" >> $O/attn64_knockouts.txt; done
  step loss probe; TAG=default python3 $R/tools/loss_probe.py > $O/loss_probe.txt 2>&1; TAG=thr0 TTV_ATTN_THR=0 python3 $R/tools/loss_probe.py >> $O/loss_probe.txt 2>&1
fi
if [ $PART = r04 ]; then       # round 4 evidence set: the driver's command, both modes under rocprof, traffic counters, base / base5, exact index
  step bench default; timeout -k 10 500 python3 $R/bench.py > $O/tiny_bench.json 2> $O/tiny_bench.err
  step bench in-flight 1; timeout -k 10 300 python3 $R/bench.py --in-flight 1 --no-cpu-baseline --no-side-legs > $O/tiny_bench_inflight1.json 2>> $O/tiny_bench.err
  step rocprof two chains; timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_if2 -o p -- python3 $R/bench.py --no-cpu-baseline --no-fp32-leg --no-side-legs > $O/tiny_bench_under_rocprof.json 2> $O/prof_if2.log
  step rocprof one chain; timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_if1 -o p -- python3 $R/bench.py --in-flight 1 --no-cpu-baseline --no-fp32-leg --no-side-legs > $O/tiny_bench_inflight1_under_rocprof.json 2> $O/prof_if1.log
  step pmc fetch; timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_f -o f -- python3 $R/bench.py --in-flight 1 --steps 20 --warmup 5 --no-cpu-baseline --no-fp32-leg --no-side-legs > /dev/null 2> $O/pmc_f.log
  step pmc write; timeout -k 10 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_w -o w -- python3 $R/bench.py --in-flight 1 --steps 20 --warmup 5 --no-cpu-baseline --no-fp32-leg --no-side-legs > /dev/null 2> $O/pmc_w.log
  step exact index; timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_s3 -o p -- python3 $R/tools/exact_index_bench.py > $O/exact_index_split3.txt 2> $O/prof_s3.log
  MODE=fp32 timeout -k 10 300 python3 $R/tools/exact_index_bench.py > $O/exact_index_fp32.txt 2>&1
  find $O -name "*kernel_trace.csv" -delete
fi
if [ $PART = r05 ]; then       # round 5 evidence set: the driver's command, both modes under rocprof, traffic counters, attention alone + stamps + clock, power probe
  step bench default; timeout -k 10 500 python3 $R/bench.py > $O/tiny_bench.json 2> $O/tiny_bench.err
  step bench in-flight 1; timeout -k 10 300 python3 $R/bench.py --in-flight 1 --no-cpu-baseline --no-side-legs > $O/tiny_bench_inflight1.json 2>> $O/tiny_bench.err
  step rocprof two chains; timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_if2 -o p -- python3 $R/bench.py --no-cpu-baseline --no-fp32-leg --no-side-legs > $O/tiny_bench_under_rocprof.json 2> $O/prof_if2.log
  step rocprof one chain; timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_if1 -o p -- python3 $R/bench.py --in-flight 1 --no-cpu-baseline --no-fp32-leg --no-side-legs > $O/tiny_bench_inflight1_under_rocprof.json 2> $O/prof_if1.log
  step pmc fetch; timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_f -o f -- python3 $R/bench.py --in-flight 1 --steps 20 --warmup 5 --no-cpu-baseline --no-fp32-leg --no-side-legs > /dev/null 2> $O/pmc_f.log
  step pmc write; timeout -k 10 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_w -o w -- python3 $R/bench.py --in-flight 1 --steps 20 --warmup 5 --no-cpu-baseline --no-fp32-leg --no-side-legs > /dev/null 2> $O/pmc_w.log
  step attention alone; FP32=0 timeout -k 10 200 python3 $R/tools/attn_bench.py 1.5 6 > $O/attn_bench.txt 2>&1
  for b in 48 64; do echo "== B=$b" >> $O/attn_bench.txt; B=$b timeout -k 10 200 python3 $R/tools/attn_bench.py 1.5 2>&1 | grep qscaled >> $O/attn_bench.txt; done
  echo "== base shape" >> $O/attn_bench.txt; B=4 CLIP=32,256,256 K=1024 HQ=12 HKV=4 FP32=0 timeout -k 10 200 python3 $R/tools/attn_bench.py 1.5 >> $O/attn_bench.txt 2>&1
  step stamps + clock; TTV_LIB_PATH=$R/titok_video_amd/csrc/build/libtitok_hip_swpstamps.so timeout -k 10 200 python3 $R/tools/swp_stamps.py > $O/swp_stamps.txt 2>&1
  echo "== all-zero operands" >> $O/swp_stamps.txt; ZERO=1 TTV_LIB_PATH=$R/titok_video_amd/csrc/build/libtitok_hip_swpstamps.so timeout -k 10 200 python3 $R/tools/swp_stamps.py >> $O/swp_stamps.txt 2>&1
  step power probe; timeout -k 10 200 python3 $R/tools/power_probe.py > $O/power_probe.txt 2>&1
  for z in 0 1; do echo "== attn_bench B=64 ZERO=$z" >> $O/power_probe.txt; ZERO=$z B=64 timeout -k 10 200 python3 $R/tools/attn_bench.py 1.5 2>&1 | grep qscaled >> $O/power_probe.txt; done
  for i in 1 2; do
    case $i in 1) C="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU";; 2) C="SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_MFMA GRBM_GUI_ACTIVE";; esac
    step sq pass $i: $C; FP32=0 timeout -k 10 300 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $O/sq$i -o s -- python3 $R/tools/attn_bench.py 1.5 > /dev/null 2> $O/sq$i.log
  done
  find $O -name "*kernel_trace.csv" -delete
fi
if [ $PART = r04base ]; then
  step base; timeout -k 10 300 python3 $R/bench.py --config base > $O/base_bench.json 2> $O/base_bench.err
  step base5; timeout -k 10 300 python3 $R/bench.py --config base5 > $O/base5_bench.json 2>> $O/base_bench.err
  step base5 decoder only; timeout -k 10 200 python3 $R/bench.py --config base5 --fp8 mx-decoder --no-cpu-baseline --no-fp32-leg > $O/base5_decoder_only_bench.json 2>> $O/base_bench.err
  step base rocprof; timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_base -o p -- python3 $R/bench.py --config base --in-flight 1 --no-cpu-baseline --no-fp32-leg > $O/base_bench_inflight1_under_rocprof.json 2> $O/prof_base.log
  step base5 rocprof; timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_base5 -o p -- python3 $R/bench.py --config base5 --in-flight 1 --no-cpu-baseline --no-fp32-leg > $O/base5_bench_inflight1_under_rocprof.json 2> $O/prof_base5.log
  find $O -name "*kernel_trace.csv" -delete
fi
if [ $PART = r04attn ]; then   # the attention kernel alone: stamps, timeline, both shapes, three SQ counter passes (diagnostic builds: tools/attn_stamps.sh, attn_timeline.sh)
  step stamps; TTV_LIB_PATH=$R/titok_video_amd/csrc/build/libtitok_hip_stamps.so timeout -k 10 200 python3 $R/tools/attn_stamps.py > $O/attn_stamps.txt 2>&1
  step timeline; TTV_LIB_PATH=$R/titok_video_amd/csrc/build/libtitok_hip_timeline.so timeout -k 10 200 python3 $R/tools/attn_timeline.py > $O/attn_timeline.txt 2>&1
  step attention alone; FP32=0 timeout -k 10 200 python3 $R/tools/attn_bench.py 1.5 6 > $O/attn_bench.txt 2>&1
  B=4 CLIP=32,256,256 K=1024 HQ=12 HKV=4 FP32=0 timeout -k 10 200 python3 $R/tools/attn_bench.py 1.5 >> $O/attn_bench.txt 2>&1
  for i in 1 2 3; do
    case $i in 1) C="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY";; 2) C="SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_MFMA";; 3) C="SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE";; esac
    step sq pass $i: $C; FP32=0 timeout -k 10 300 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $O/sq$i -o s -- python3 $R/tools/attn_bench.py 1.5 > /dev/null 2> $O/sq$i.log
  done
fi
if [ $PART = tiny ]; then
  step bench default; python3 $R/bench.py > $O/tiny_bench.json 2> $O/tiny_bench.err
  step bench in-flight 1; python3 $R/bench.py --in-flight 1 --no-cpu-baseline > $O/tiny_bench_inflight1.json 2>> $O/tiny_bench.err
  step rocprof two chains; rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_if2 -o p -- python3 $R/bench.py --no-cpu-baseline --no-fp32-leg > $O/tiny_bench_under_rocprof.json 2> $O/prof_if2.log
  step rocprof one chain; rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_if1 -o p -- python3 $R/bench.py --in-flight 1 --no-cpu-baseline --no-fp32-leg > $O/tiny_bench_inflight1_under_rocprof.json 2> $O/prof_if1.log
  step ubench; $R/tools/ubench/valu_rates > $O/ubench_valu_rates.txt 2>&1
  step stamps; TTV_LIB_PATH=$R/titok_video_amd/csrc/build/libtitok_hip_stamps.so python3 $R/tools/attn_stamps.py > $O/attn_stamps.txt 2>&1
  TTV_LIB_PATH=$R/titok_video_amd/csrc/build/libtitok_hip_stamps64.so python3 $R/tools/attn64_stamps.py > $O/attn64_stamps.txt 2>&1
  step attention alone; FP32=0 python3 $R/tools/attn_bench.py 1.5 6 > $O/attn_bench.txt 2>&1
  B=4 CLIP=32,256,256 K=1024 HQ=12 HKV=4 FP32=0 python3 $R/tools/attn_bench.py 1.5 >> $O/attn_bench.txt 2>&1
fi
if [ $PART = counters ]; then
  step pmc fetch; rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_f -o f -- python3 $R/bench.py --in-flight 1 --steps 20 --warmup 5 --no-cpu-baseline --no-fp32-leg > /dev/null 2> $O/pmc_f.log
  step pmc write; rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_w -o w -- python3 $R/bench.py --in-flight 1 --steps 20 --warmup 5 --no-cpu-baseline --no-fp32-leg > /dev/null 2> $O/pmc_w.log
  for i in 1 2 3; do
    case $i in 1) C="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY";; 2) C="SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_MFMA";; 3) C="SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE";; esac
    step sq pass $i: $C; FP32=0 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $O/sq$i -o s -- python3 $R/tools/attn_bench.py 1.5 > /dev/null 2> $O/sq$i.log
  done
fi
if [ $PART = base ]; then
  step base bench l2; python3 $R/bench.py --config base > $O/base_bench.json 2> $O/base_bench.err
  step base bench fsq; python3 $R/bench.py --config base --quantizer fsq --no-cpu-baseline --no-fp32-leg > $O/base_fsq_bench.json 2>> $O/base_bench.err
  step base fp8; python3 $R/bench.py --config base --quantizer fsq --fp8 --no-cpu-baseline --no-fp32-leg > $O/base_fp8_bench.json 2>> $O/base_bench.err
  step base rocprof; rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_base -o p -- python3 $R/bench.py --config base --no-cpu-baseline --no-fp32-leg > $O/base_bench_under_rocprof.json 2> $O/prof_base.log
fi
if [ $PART = train ]; then
  step train bench; python3 $R/tools/bench_train.py > $O/bench_train.txt 2>&1
  B=5 python3 $R/tools/bench_train.py >> $O/bench_train.txt 2>&1
  step train rocprof; rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_train -o p -- python3 $R/tools/bench_train.py > $O/bench_train_under_rocprof.txt 2> $O/prof_train.log
  step config 3; python3 $R/tools/train_dp.py --steps 100 --warmup 10 > $O/train_dp.json 2> $O/train_dp.err
  python3 $R/tools/train_dp.py --gpus 2 --backend gloo --steps 40 --warmup 5 > $O/train_dp_2ranks_gloo.json 2>> $O/train_dp.err
  B=5 python3 $R/tools/bench_gan_train.py > $O/bench_gan_train.txt 2>&1
  python3 $R/tools/loader_bench.py > $O/loader_bench.txt 2>&1
fi
echo "[collect $PART] done"
