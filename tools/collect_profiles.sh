#!/bin/bash
# Evidence for profiles/<round>/ : run on the GPU box (gpurun), writes gpurun_out/<round>/.
# usage: tools/collect_profiles.sh r01
R=${1:-r01}
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/$R
mkdir -p $O
python -m pytest tests -q -m gpu 2>&1 | grep -E "passed|failed|error" | tail -3 > $O/pytest_gpu.txt
python bench.py                                   | tail -1 > $O/bench_distill_n1.json
python bench.py --serial --no-cpu-baseline        | tail -1 > $O/bench_distill_serial_n1.json
python bench.py --workload student --no-cpu-baseline | tail -1 > $O/bench_student_n1.json
python bench.py --workload teacher --no-cpu-baseline | tail -1 > $O/bench_teacher_n1.json
python bench.py --workload joint --steps 10 --no-cpu-baseline | tail -1 > $O/bench_joint_n1.json
python bench.py --frames 13 --steps 10 --no-cpu-baseline --teacher senet50 | tail -1 > $O/bench_distill_13frames_senet50_n1.json
# per-kernel durations of the serial pass (what roofline.avg_launch_ms is compared with)
timeout 300 rocprofv3 --kernel-trace --stats -d $O/kt -o run -- python bench.py --serial --no-cpu-baseline \
    > $O/bench_under_rocprof.json 2> $O/rocprof_kt.log
DB=$(find $O/kt -name "*.db" | head -1)
python tools/prof_summary.py "$DB" 44 > $O/kernel_stats.txt   # 3 warm-up + 20 timed + 1 + 20 roofline-leg steps
rm -rf $O/kt
# PMC passes, each on its own (no tracing domains besides the kernel trace)
SQC="SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA"
timeout 300 rocprofv3 --kernel-trace --pmc $SQC --output-format csv -d $O/pmc_sq -- python bench.py --serial --steps 3 --warmup 3 --no-cpu-baseline --no-roofline > /dev/null 2> $O/rocprof_sq.log
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -- python bench.py --serial --steps 3 --warmup 3 --no-cpu-baseline --no-roofline > /dev/null 2> $O/rocprof_fetch.log
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -- python bench.py --serial --steps 3 --warmup 3 --no-cpu-baseline --no-roofline > /dev/null 2> $O/rocprof_write.log
python tools/pmc_table.py $O/pmc_sq $O/pmc_fetch $O/pmc_write $O/pmc_summary.txt $O/pmc_traffic.json
rm -rf $O/pmc_sq $O/pmc_fetch $O/pmc_write $O/rocprof_*.log
ls -la $O
