#!/bin/bash
# Evidence for profiles/<round>/ : run on the GPU box (gpurun), writes gpurun_out/<round>/.
# usage: tools/collect_profiles.sh r02 [commit]
R=${1:-r06}
COMMIT=${2:-unknown}
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/$R
mkdir -p $O
B="python bench.py"
( time python -m pytest tests -q -m gpu --durations=8 ) 2>&1 | grep -E "passed|failed|error|^real|s call" | tail -14 > $O/pytest_gpu.txt
python __graft_entry__.py smoke 2>&1 | grep smoke >> $O/pytest_gpu.txt
$B                                                          | tail -1 > $O/bench_distill_n1.json
$B --steps 20 --warmup 5 --no-cpu-baseline                  | tail -1 > $O/bench_distill_driver_flags_n1.json   # the driver's command line
$B --serial --exec-hint 0 --no-cpu-baseline                 | tail -1 > $O/bench_distill_serial_n1.json   # the timed region's kernels on one stream
$B --workload student --no-cpu-baseline                     | tail -1 > $O/bench_student_n1.json
$B --workload teacher --no-cpu-baseline                     | tail -1 > $O/bench_teacher_n1.json
$B --workload joint --no-cpu-baseline                       | tail -1 > $O/bench_joint_n1.json
$B --teacher senet50 --no-cpu-baseline                      | tail -1 > $O/bench_distill_senet50_n1.json          # run_distillation.m:82 default teacher, F = 1
$B --teacher senet50 --per-gpu-batch 256 --no-cpu-baseline  | tail -1 > $O/bench_distill_senet50_b256_n1.json     # north_star: batch 256 on one GPU
$B --per-gpu-batch 256 --no-cpu-baseline                    | tail -1 > $O/bench_distill_b256_n1.json
$B --frames 13 --no-cpu-baseline --teacher senet50          | tail -1 > $O/bench_distill_13frames_senet50_n1.json
$B --imdb-windows 1 --no-cpu-baseline --teacher senet50     | tail -1 > $O/bench_distill_imdb_windows_senet50_n1.json
$B --workload cpu-teacher                                   | tail -1 > $O/bench_cpu_teacher.json                  # BASELINE config 1 (host cores only)
XM_DEBUG_DIST=1 $B --parserv rccl-capi --no-cpu-baseline --no-roofline --north-star 0 | tail -1 > $O/bench_distill_capi_1rank.json
XM_DEBUG_DIST=1 $B --parserv torch --no-cpu-baseline --no-roofline --north-star 0 | tail -1 > $O/bench_distill_torch_1rank.json
XM_DEBUG_DIST=1 XM_PS_LATE=1 $B --parserv rccl-capi --no-cpu-baseline --no-roofline --north-star 0 | tail -1 > $O/bench_distill_capi_late_init.json   # communicator created AFTER the nets: the 12 % trap (xmodal.h CALL ORDER)
# round 4: `bench.py --gpus 2` with no launcher around it starts its two ranks itself (XM_DEBUG_DIST=gloo0: both on this box's one
# GPU, exchange over gloo -- a functional run of the N > 1 path; the throughput means nothing)
XM_DEBUG_DIST=gloo0 $B --gpus 2 --steps 5 --warmup 2 --no-cpu-baseline --no-roofline 2>/dev/null | grep "^{" | tail -1 > $O/bench_distill_gpus2_gloo0.json
# round 5: the reference's real default shape (numSeconds = 4 -> 512 x 400 spectrograms, run_distillation.m:74)
$B --workload student --width 400 --no-cpu-baseline        | tail -1 > $O/bench_student_w400_n1.json
$B --teacher senet50 --width 400 --no-cpu-baseline --north-star 0 | tail -1 > $O/bench_distill_senet50_w400_n1.json
# one stream + the host's XM_EXEC_SINGLE_STREAM hint (what a MATLAB / MEX host runs) at the north_star batch
$B --serial --teacher senet50 --per-gpu-batch 256 --no-cpu-baseline --no-roofline | tail -1 > $O/bench_distill_senet50_b256_serial_hint1.json
# (the schedule / priority / variant A/Bs of rounds 3-4 are not repeated: profiles/r03, profiles/r04 hold them; this
# round's kernel A/Bs -- run inside the round, each in one gpurun call -- are the *_bench.txt / dma_kernel_dissection.txt files)
# round 6: the student's conv1 -> bn1 -> relu1 -> pool1 chain, composed operators against the Gram route
for n in 32 64 256; do python tools/stem_chain_bench.py $n 2>&1 | grep -v amdgpu.ids; done > $O/stem_chain_bench.txt
XM_NO_STEM_FWD=1 XM_NO_STEM_GRAM=1 $B --teacher senet50 --per-gpu-batch 256 --no-cpu-baseline --no-roofline | tail -1 > $O/bench_distill_senet50_b256_round5_stem.json   # the other arm: round-5 stem kernels
XM_NO_STEM_FWD=1 XM_NO_STEM_GRAM=1 $B --no-cpu-baseline --no-roofline --north-star 0 | tail -1 > $O/bench_distill_round5_stem.json
# per-kernel durations of the serial pass (what roofline.avg_launch_ms is compared with)
timeout 600 rocprofv3 --kernel-trace --stats -d $O/kt -o run -- $B --serial --exec-hint 0 --no-cpu-baseline --steps 60 --warmup 10 \
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
  timeout 600 rocprofv3 --kernel-trace --stats -d $O/kt_$tag -o run -- $B --serial --exec-hint 0 --no-cpu-baseline --no-roofline --steps 20 --warmup 5 "$@" \
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
P="--serial --exec-hint 0 --steps 3 --warmup 3 --no-cpu-baseline --no-roofline"
SQC="SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA"
timeout 600 rocprofv3 --kernel-trace --pmc $SQC --output-format csv -d $O/pmc_sq -- $B $P > /dev/null 2> $O/rocprof_sq.log
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -- $B $P > /dev/null 2> $O/rocprof_fetch.log
timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -- $B $P > /dev/null 2> $O/rocprof_write.log
python tools/pmc_table.py $O/pmc_sq $O/pmc_fetch $O/pmc_write $O/pmc_summary.txt $O/pmc_traffic.json $COMMIT
rm -rf $O/pmc_sq $O/pmc_fetch $O/pmc_write $O/rocprof_*.log
# round 5: the same three passes on north_star's batch (where conv_dgrad_s2_kernel / the large-launch patch kernels run)
P2="$P --teacher senet50 --per-gpu-batch 256 --steps 2 --warmup 2"
timeout 900 rocprofv3 --kernel-trace --pmc $SQC --output-format csv -d $O/pmc_sq -- $B $P2 > /dev/null 2> $O/rocprof_sq.log
timeout 900 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -- $B $P2 > /dev/null 2> $O/rocprof_fetch.log
timeout 900 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -- $B $P2 > /dev/null 2> $O/rocprof_write.log
python tools/pmc_table.py $O/pmc_sq $O/pmc_fetch $O/pmc_write $O/pmc_summary_senet50_b256.txt $O/pmc_traffic_senet50_b256.json $COMMIT
rm -rf $O/pmc_sq $O/pmc_fetch $O/pmc_write $O/rocprof_*.log
ls -la $O
