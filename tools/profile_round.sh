#!/bin/bash
# Collect the round's measurement evidence on a GPU box (run through gpurun from the repo root):
#   bench JSON lines (fp32 + int8, PARITY + FAST, single stream, multi-rank rehearsal), rocprofv3 kernel stats, HBM traffic from
#   PMC counters (separate passes), SQ / LDS counters, in-kernel phase clocks, the reference demo's real-time factors.
# Everything lands in gpurun_out/prof/; tools/profile_summarize.py <round> turns it into profiles/<round>_*.
set -u
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
RND=${1:-r06}
OUT=gpurun_out/prof
rm -rf $OUT; mkdir -p $OUT
# kernel stats + HBM traffic first: bench.py quotes `roofline.traffic` from profiles/<round>_hbm_traffic*.json, which must have been
# measured with these very kernel sources (kernel_source_hash)
# (LPCNET_HIP_NO_AUTOTUNE=1: only the workload's launches in the traces -- the auto-tune's short trial launches of every streams-per-workgroup
# variant would be averaged into the kernel statistics; the table value is what the tuned bench line chooses on this model, see `streams_per_workgroup`)
for fl in f32 i8 f32_fast i8_fast; do
  case $fl in f32) flag="";; i8) flag="--int8";; f32_fast) flag="--fast";; i8_fast) flag="--int8 --fast --spw 2";; esac
  if [ $fl = f32 ] || [ $fl = i8 ]; then
    LPCNET_HIP_NO_AUTOTUNE=1 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_$fl -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras $flag > $OUT/stats_$fl.log 2>&1
  fi
  LPCNET_HIP_NO_AUTOTUNE=1 timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/fetch_$fl -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras $flag > $OUT/fetch_$fl.log 2>&1
  LPCNET_HIP_NO_AUTOTUNE=1 timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/write_$fl -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras $flag > $OUT/write_$fl.log 2>&1
done
python tools/profile_summarize.py $RND > /dev/null 2>&1
timeout 600 python bench.py > $OUT/bench_f32.json 2> $OUT/bench_f32.err
timeout 300 python bench.py --int8 > $OUT/bench_i8.json 2> $OUT/bench_i8.err
timeout 300 python bench.py --streams 1 --no-cpu-baseline > $OUT/bench_single.json 2> $OUT/bench_single.err      # BASELINE config 1
# the operating point the metric is named after: one frame per step for every stream against the 10-ms deadline, >= 500 steps
# (round 6, VERDICT r5 item 3: 2000 consecutive steps at the sustained counts, wall AND device time of every step)
timeout 900 python bench.py --rt --steps 2000 --rt-sweep 1024,6144,7168,8192 --no-cpu-baseline > $OUT/bench_rt_f32.json 2> $OUT/bench_rt_f32.err
timeout 900 python bench.py --rt --steps 2000 --rt-sweep 1024,9216,9728,10240,10752 --no-cpu-baseline --int8 > $OUT/bench_rt_int8.json 2> $OUT/bench_rt_int8.err
# trained-like (heavy-tailed) GRU-A sparsity: the 80-item variants, items past the 28th and their block indices streamed from L2
timeout 300 python bench.py --skew 0.1 --no-cpu-baseline > $OUT/bench_skewed.json 2> $OUT/bench_skewed.err
timeout 300 python bench.py --skew 0.1 --int8 --no-cpu-baseline > $OUT/bench_skewed_int8.json 2> $OUT/bench_skewed_int8.err
# denser GRU-A models: the variants that stream their items past the 28th from L2
for dn in nw32:0.06,0.06,0.22 nw36:0.07,0.07,0.25 nw40:0.08,0.08,0.3 nw48:0.1,0.1,0.35; do
  timeout 300 python bench.py --densities ${dn#*:} --no-cpu-baseline > $OUT/bench_denseA_${dn%%:*}.json 2> $OUT/bench_denseA_${dn%%:*}.err
done
# throughput at other batch sizes
for ns in 256 512 1024 1536 4096 8192; do
  timeout 300 python bench.py --streams $ns --no-cpu-baseline --steps 6 --warmup 2 > $OUT/bench_f32_n$ns.json 2> $OUT/bench_f32_n$ns.err
  timeout 300 python bench.py --streams $ns --no-cpu-baseline --steps 6 --warmup 2 --int8 > $OUT/bench_i8_n$ns.json 2> $OUT/bench_i8_n$ns.err
done
timeout 300 python bench.py --fast --no-cpu-baseline > $OUT/bench_f32_fast.json 2> $OUT/bench_f32_fast.err
timeout 300 python bench.py --fast --fp16-fc --no-cpu-baseline > $OUT/bench_f32_fast_f16.json 2> $OUT/bench_f32_fast_f16.err
timeout 300 python bench.py --int8 --fast --spw 2 --no-cpu-baseline > $OUT/bench_i8_fast.json 2> $OUT/bench_i8_fast.err
timeout 300 python bench.py --int8 --fast --fp16-fc --spw 2 --no-cpu-baseline > $OUT/bench_i8_fast_f16.json 2> $OUT/bench_i8_fast_f16.err   # BASELINE config 4 as worded
timeout 600 python bench.py --gpus 2 --share-device --no-cpu-baseline > $OUT/bench_rehearsal_2ranks.json 2> $OUT/bench_rehearsal_2ranks.err
# BASELINE configs 3 / 4 at full shape on the one GPU there is: 8 ranks x 1024 streams (a plumbing rehearsal, not a scaling number)
timeout 900 python bench.py --gpus 8 --share-device --steps 3 --warmup 1 --no-cpu-baseline > $OUT/bench_rehearsal_8ranks.json 2> $OUT/bench_rehearsal_8ranks.err
timeout 900 python bench.py --gpus 8 --share-device --steps 3 --warmup 1 --no-cpu-baseline --int8 > $OUT/bench_rehearsal_8ranks_int8.json 2> $OUT/bench_rehearsal_8ranks_int8.err
# BASELINE configs 0 -> 1: the reference's own demo on the engine vs its AVX2 builds, one 10-s feature file (wall seconds incl. process start)
python tools/rtf_demo.py > $OUT/rtf_demo.json 2> $OUT/rtf_demo.err
python tools/profile_sq.py --tag $RND > $OUT/sq_f32.log 2>&1
python tools/profile_sq.py --tag ${RND}_n1024 --extra="--streams 1024" > $OUT/sq_f32_n1024.log 2>&1
python tools/profile_sq.py --tag $RND --extra=--fast > $OUT/sq_f32_fast.log 2>&1
python tools/profile_sq.py --int8 --tag $RND > $OUT/sq_i8.log 2>&1
# in-kernel s_memtime phase tables (profiling build of the library: LPCN_PROF_MASK=0xFFF python -m lpcnet_amd.build --prof)
if [ -f lpcnet_amd/liblpcnet_hip_prof.so ]; then
  LPCNET_HIP_LIB=$PWD/lpcnet_amd/liblpcnet_hip_prof.so python tests/tools/x2_phase.py 12 2048 > $OUT/phase_x2.log 2>&1
  LPCNET_HIP_LIB=$PWD/lpcnet_amd/liblpcnet_hip_prof.so python tests/tools/gpu_sweep.py 22 1024:4,1:1 > $OUT/phase_f32.log 2>&1
  LPCN_FLAVOUR=int8 LPCNET_HIP_LIB=$PWD/lpcnet_amd/liblpcnet_hip_prof.so python tests/tools/gpu_sweep.py 22 1024:4,1024:2 > $OUT/phase_i8.log 2>&1
  LPCN_FAST=1 LPCNET_HIP_LIB=$PWD/lpcnet_amd/liblpcnet_hip_prof.so python tests/tools/gpu_sweep.py 22 1024:4 > $OUT/phase_f32_fast.log 2>&1
  LPCN_FAST=1 LPCN_FLAVOUR=int8 LPCNET_HIP_LIB=$PWD/lpcnet_amd/liblpcnet_hip_prof.so python tests/tools/gpu_sweep.py 22 1024:2 > $OUT/phase_i8_fast.log 2>&1
fi
find $OUT -name "*.csv" | head -40
