#!/bin/bash
# Evidence for profiles/<round>/ : run on the GPU box (gpurun), writes gpurun_out/<round>/.
# usage: tools/collect_profiles.sh r02 [commit]
R=${1:-r04}
COMMIT=${2:-unknown}
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/$R
mkdir -p $O
B="python bench.py"
python -m pytest tests -q -m gpu 2>&1 | grep -E "passed|failed|error" | tail -3 > $O/pytest_gpu.txt
$B                                                          | tail -1 > $O/bench_distill_n1.json
$B --steps 20 --warmup 5 --no-cpu-baseline                  | tail -1 > $O/bench_distill_driver_flags_n1.json   # the driver's command line
$B --serial --no-cpu-baseline                               | tail -1 > $O/bench_distill_serial_n1.json
$B --workload student --no-cpu-baseline                     | tail -1 > $O/bench_student_n1.json
$B --workload teacher --no-cpu-baseline                     | tail -1 > $O/bench_teacher_n1.json
$B --workload joint --no-cpu-baseline                       | tail -1 > $O/bench_joint_n1.json
$B --teacher senet50 --no-cpu-baseline                      | tail -1 > $O/bench_distill_senet50_n1.json          # run_distillation.m:82 default teacher, F = 1
$B --teacher senet50 --per-gpu-batch 256 --no-cpu-baseline  | tail -1 > $O/bench_distill_senet50_b256_n1.json     # north_star: batch 256 on one GPU
$B --per-gpu-batch 256 --no-cpu-baseline                    | tail -1 > $O/bench_distill_b256_n1.json
$B --frames 13 --no-cpu-baseline --teacher senet50          | tail -1 > $O/bench_distill_13frames_senet50_n1.json
$B --workload cpu-teacher                                   | tail -1 > $O/bench_cpu_teacher.json                  # BASELINE config 1 (host cores only)
XM_DEBUG_DIST=1 $B --parserv rccl-capi --no-cpu-baseline --no-roofline --north-star 0 | tail -1 > $O/bench_distill_capi_1rank.json
XM_DEBUG_DIST=1 $B --parserv torch --no-cpu-baseline --no-roofline --north-star 0 | tail -1 > $O/bench_distill_torch_1rank.json
XM_DEBUG_DIST=1 XM_PS_LATE=1 $B --parserv rccl-capi --no-cpu-baseline --no-roofline --north-star 0 | tail -1 > $O/bench_distill_capi_late_init.json   # communicator created AFTER the nets: the 12 % trap (xmodal.h CALL ORDER)
# round 4: `bench.py --gpus 2` with no launcher around it starts its two ranks itself (XM_DEBUG_DIST=gloo0: both on this box's one
# GPU, exchange over gloo -- a functional run of the N > 1 path; the throughput means nothing)
XM_DEBUG_DIST=gloo0 $B --gpus 2 --steps 5 --warmup 2 --no-cpu-baseline --no-roofline 2>/dev/null | grep "^{" | tail -1 > $O/bench_distill_gpus2_gloo0.json
# round-3 experiments behind DESIGN.md 2.3b: where the main stream spends its time; scheduling variants that moved nothing
NS="--no-cpu-baseline --no-roofline --north-star 0"
{
  for f in "" "--wgrad-stream 0" "--overlap-teacher 0" "--serial" "--workload student --per-gpu-batch 32" "--workload student --per-gpu-batch 32 --wgrad-stream 0"; do
    echo "== bench.py $f"; XM_BENCH_MARKS=1 $B $NS $f 2>&1 | grep -E "marks|\"value\"" | sed -e "s/, \"unit.*//"; done
} > $O/phase_marks.txt
{
  for tb in 0 64 128 256; do echo -n "--teacher-batch $tb: "; $B $NS --teacher-batch $tb 2>/dev/null | tail -1 | cut -c60-112; done
  for g in "" loss conv5 conv3 conv2 bn1; do echo -n "--teacher-gate '$g': "; $B $NS --teacher-gate "$g" 2>/dev/null | tail -1 | cut -c60-112; done
  for e in "XM_X=1" "XM_NO_FUSED_STEM_BWD=1" "XM_WGRAD_AFTER_DGRAD=1" "XM_SIDE_PRIO=0 XM_MAIN_PRIO=-1" "XM_NO_FUSED_STATS=1" "XM_NO_FUSED_BIASDER=1" "XM_NO_FAST_TRANSPOSE=1" "XM_NO_HALO=1" "XM_NO_HYBRID=1" "XM_NO_STEM=1" "XM_NO_STEM_WGRAD=1" "XM_NO_SKINNY4=1" "XM_TUNE_FILE= XM_HALO_MARGIN=0.04" "XM_TUNE_FILE= XM_HALO_MARGIN=0.015"; do
    echo -n "$e: "; env $e $B $NS 2>/dev/null | tail -1 | cut -c60-112; done
  # round 4: the eight-wave configuration (conv.hip kCfgs[7]) inside the overlapped step: never / for launches of >= 1024 tiles
  # (shipped) / always.  The shipped table was tuned with the default; the other two lines tune in the process (XM_TUNE_FILE=).
  for w in "default (>= 1024 tiles), shipped table|XM_X=1" "never|XM_TUNE_FILE= XM_NO_W8=1" "default, tuned in the process|XM_TUNE_FILE= XM_X=1" "always|XM_TUNE_FILE= XM_W8_MIN_TILES=0"; do
    for wl in "" "--per-gpu-batch 256 --steps 10 --warmup 3" "--workload student" "--workload joint"; do
      echo -n "eight-wave configuration ${w%%|*}, bench.py $wl: "; env ${w##*|} $B $NS $wl 2>/dev/null | tail -1 | grep -o '"value": [0-9.]*, "unit": "[a-z/]*"'; done; done
  # round 4: conv_wgrad_patch_kernel inside the step: one stream (where it is taken) and two streams (where it is not, and forced)
  for e in "XM_X=1" "XM_NO_WGRAD_PATCH=1"; do for wl in "--serial" "--workload student --wgrad-stream 0" "--serial --per-gpu-batch 256 --steps 10 --warmup 3"; do
    echo -n "one stream, $e, bench.py $wl: "; env $e $B $NS $wl 2>/dev/null | tail -1 | grep -o '"value": [0-9.]*, "unit": "[a-z/]*"'; done; done
  for e in "XM_X=1" "XM_WGRAD_PATCH_ANY_STREAM=1" "XM_WGRAD_PATCH_ANY_STREAM=1 XM_WGRAD_PATCH_SLOTS=384"; do for wl in "" "--workload student" "--per-gpu-batch 256 --steps 10 --warmup 3"; do
    echo -n "two streams, $e, bench.py $wl: "; env $e $B $NS $wl 2>/dev/null | tail -1 | grep -o '"value": [0-9.]*, "unit": "[a-z/]*"'; done; done
  for e in "XM_X=1" "XM_NO_HALO=1" "XM_NO_FUSED_STATS=1" "XM_NO_STEM=1" "XM_NO_STEM_WGRAD=1"; do echo -n "student batch 64, $e: "; env $e $B $NS --workload student 2>/dev/null | tail -1 | cut -c50-100; done
  for e in "XM_X=1" "XM_NO_FUSED_SE=1" "XM_NO_HYBRID=1"; do echo -n "config 3 (SE-ResNet50 fwd, 128), $e: "; env $e $B $NS --workload teacher 2>/dev/null | tail -1 | grep -o '"value": [0-9.]*, "unit": "[a-z/]*"'; done
  for e in "XM_X=1" "XM_NO_FUSED_SE=1" "XM_NO_HYBRID=1" "XM_NO_FUSED_STEM_BWD=1" "XM_TUNE_FILE= XM_HALO_MARGIN=0.04" "XM_TUNE_FILE= XM_HALO_MARGIN=0.015"; do echo -n "north_star batch 256 (SE-ResNet50), $e: "; env $e $B $NS --teacher senet50 --per-gpu-batch 256 --steps 10 --warmup 3 2>/dev/null | tail -1 | cut -c60-112; done
  for e in "XM_X=1" "XM_NO_FUSED_SE_BWD=1" "XM_NO_FUSED_STEM_BWD=1"; do echo -n "config-5 shard (joint, 64 pairs), $e: "; env $e $B $NS --workload joint 2>/dev/null | tail -1 | cut -c60-112; done
  for c in 32 64 128; do echo -n "north_star batch 256, frozen teacher in slices of $c faces (--teacher-chunk): "; $B $NS --teacher senet50 --per-gpu-batch 256 --steps 10 --warmup 3 --teacher-chunk $c 2>/dev/null | tail -1 | cut -c60-112; done
} > $O/schedule_experiments.txt
for n in 32 64 256; do python tools/halo_bench.py $n 2>&1 | grep -v amdgpu; done > $O/halo_bench.txt
{ for n in 32 64; do python tools/stem_bench.py $n 2>&1 | grep -v amdgpu; done; hipcc --offload-arch=gfx950 -O3 tools/store_mfma_probe.hip -o /tmp/smp 2>/dev/null && /tmp/smp; } > $O/stem_bench.txt
# round 4: configuration 0 (four waves of 222 VGPRs) against configuration 7 (eight waves of 128) on an idle device
{ for n in 32 64 256; do for c in 0 7; do echo "== $n samples, forced configuration $c"; python tools/conv_bench.py --cfg $c --n $n --reps 30 --dirs fwd,dgrad s_conv2 s_conv3 s_conv4 s_conv5 t_res3_3x3 t_res4_3x3 x_fill3x3 x_fill1x1 2>&1 | grep -v amdgpu; done; done; } > $O/w8_bench.txt
# round 4: filter derivative of the student's 3 x 3 layers, generic kernel against conv_wgrad_patch_kernel, idle device
{ for n in 32 64 256; do for e in "XM_NO_WGRAD_PATCH=1" "XM_X=1"; do echo "== $n spectrograms, $e"; env XM_TUNE_FILE= $e python tools/conv_bench.py --n $n --reps 30 --dirs wgrad s_conv3 s_conv4 s_conv5 2>&1 | grep -v amdgpu; done; done; } > $O/wgrad_patch_bench.txt
python tools/stats_bench.py 32 2>&1 | grep -v amdgpu > $O/stats_bench.txt
python tools/bnbwd_bench.py 32 2>&1 | grep -v amdgpu > $O/bnbwd_bench.txt
{ for n in 32 64 256; do python tools/stem_bwd_bench.py $n 2>&1 | grep -v amdgpu; done; python tools/mall_chunk_bench.py 32 2>&1 | grep -v amdgpu; } > $O/stem_bwd_bench.txt
# per-kernel durations of the serial pass (what roofline.avg_launch_ms is compared with)
timeout 600 rocprofv3 --kernel-trace --stats -d $O/kt -o run -- $B --serial --no-cpu-baseline --steps 60 --warmup 10 \
    > $O/bench_under_rocprof.json 2> $O/rocprof_kt.log
DB=$(find $O/kt -name "*.db" | head -1)
STEPS=$(python - <<PY
import json
d = json.loads(open("$O/bench_under_rocprof.json").read().strip().splitlines()[-1])
print(d["warmup"] + d["settle_steps"] + d["steps"] + 1 + 10 + d["roofline"]["steps"])
PY
)
python tools/prof_summary.py "$DB" $STEPS > $O/kernel_stats.txt
rm -rf $O/kt
# the same summary for the other configurations (serial mode, 20 timed steps each)
kstats() {   # kstats <tag> <bench flags...>
  local tag=$1; shift
  timeout 600 rocprofv3 --kernel-trace --stats -d $O/kt_$tag -o run -- $B --serial --no-cpu-baseline --no-roofline --steps 20 --warmup 5 "$@" \
      > $O/kt_$tag.json 2> /dev/null
  local db=$(find $O/kt_$tag -name "*.db" | head -1)
  local n=$(python -c "import json; d=json.loads(open('$O/kt_$tag.json').read().strip().splitlines()[-1]); print(d['warmup'] + d['settle_steps'] + d['steps'])")
  python tools/prof_summary.py "$db" $n > $O/kernel_stats_$tag.txt
  rm -rf $O/kt_$tag $O/kt_$tag.json
}
if [ -z "$XM_PROFILE_SKIP_EXTRA" ]; then
  kstats student --workload student
  kstats teacher --workload teacher
  kstats joint --workload joint
  kstats senet50 --teacher senet50
  kstats b256 --per-gpu-batch 256
  kstats senet50_b256 --teacher senet50 --per-gpu-batch 256
fi
# PMC passes, each on its own (no tracing domains besides the kernel trace)
P="--serial --steps 3 --warmup 3 --no-cpu-baseline --no-roofline"
SQC="SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA"
timeout 600 rocprofv3 --kernel-trace --pmc $SQC --output-format csv -d $O/pmc_sq -- $B $P > /dev/null 2> $O/rocprof_sq.log
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -- $B $P > /dev/null 2> $O/rocprof_fetch.log
timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -- $B $P > /dev/null 2> $O/rocprof_write.log
python tools/pmc_table.py $O/pmc_sq $O/pmc_fetch $O/pmc_write $O/pmc_summary.txt $O/pmc_traffic.json $COMMIT
rm -rf $O/pmc_sq $O/pmc_fetch $O/pmc_write $O/rocprof_*.log
ls -la $O
