#!/usr/bin/env python
"""Regenerates profiles/README.md from the JSON / text evidence under profiles/<round>/.
usage: python tools/make_profiles_readme.py r02"""
import json, os, re, sys
R = sys.argv[1] if len(sys.argv) > 1 else "r05"
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P = os.path.join(root, "profiles", R)
def L(n): return json.loads(open(os.path.join(P, n + ".json")).read().strip().splitlines()[-1])
d, dd, ds, st, te, jo, mf = (L("bench_distill_n1"), L("bench_distill_driver_flags_n1"), L("bench_distill_serial_n1"),
                             L("bench_student_n1"), L("bench_teacher_n1"), L("bench_joint_n1"),
                             L("bench_distill_13frames_senet50_n1"))
se, se256, r256, c1, capi = (L("bench_distill_senet50_n1"), L("bench_distill_senet50_b256_n1"), L("bench_distill_b256_n1"),
                             L("bench_cpu_teacher"), L("bench_distill_capi_1rank"))
tor1, late = L("bench_distill_torch_1rank"), L("bench_distill_capi_late_init")
w400, dw400, sh1, g2 = (L("bench_student_w400_n1"), L("bench_distill_senet50_w400_n1"), L("bench_distill_senet50_b256_serial_hint1"),
                        L("bench_distill_gpus2_gloo0"))
tb = json.load(open(os.path.join(P, "pmc_traffic_senet50_b256.json")))
ns = d.get("north_star_b256") or {}
r, c = d["roofline"], d["cpu_baseline"]
tests = open(os.path.join(P, "pytest_gpu.txt")).read().strip().splitlines()[-1]
ks = open(os.path.join(P, "kernel_stats.txt")).read()
m = re.search(re.escape(r["kernel"]) + r".*?\s(\d+)\s+([\d.]+)\s+([\d.]+)\s+[\d.]+\n", ks)
rocavg = float(m.group(3)) if m else float("nan")
tot = re.search(r"TOTAL kernel time: .*", ks).group(0)
meta = json.load(open(os.path.join(P, "pmc_traffic.json"))).get("_meta", {})
pct = lambda x: "%.1f %%" % (100 * x)
w = d.get("windows") or {}
rows = "\n".join("| `%s` | %.3f | %.1f | %d |" % (k["kernel"], k["ms_per_step"], k["tflops"], k["launches_per_step"])
                 for k in r["per_kernel"])
txt = f"""# profiles/ — measured evidence, round {int(R[1:])} (MI355X, 1 GPU, ROCm 7.2)

Everything under `{R}/` (except the files marked "own gpurun call" below) comes from ONE `gpurun` call on a fresh MI355X box at commit `{meta.get('commit')}`:
`bash tools/collect_profiles.sh {R} <commit>` (the script lists every command); this file is generated from
those files by `tools/make_profiles_readme.py {R}`.  `r01/` ... `r04/` are the previous rounds' evidence, unchanged (the schedule /
priority / variant A/Bs of rounds 3-4 -- `r03/schedule_experiments.txt`, `r04/schedule_experiments.txt`, `phase_marks.txt`, `w8_bench.txt`,
`halo_bench.txt`, `stem_bench.txt` ... -- were not repeated: no kernel they cover changed).

| file | what |
|---|---|
| `{R}/pytest_gpu.txt` | `python -m pytest tests -q -m gpu` on that box ({tests}) |
| `{R}/bench_distill_n1.json` | `python bench.py` — default line: distillation step, 32 pairs/GPU, K chosen for a >= 2 s timed region, incl. `roofline`, `cpu_baseline`, `windows` |
| `{R}/bench_distill_driver_flags_n1.json` | `python bench.py --steps 20 --warmup 5` — the driver's command line (settle phase in front of the 20 timed steps: `settle_steps` in the line) |
| `{R}/bench_distill_serial_n1.json` | `python bench.py --serial --exec-hint 0` — same step, same kernels, on ONE HIP stream (no overlap): the mode the roofline leg and the profiles below use |
| `{R}/bench_under_rocprof.json`, `{R}/kernel_stats.txt` | `rocprofv3 --kernel-trace --stats -- python bench.py --serial --exec-hint 0 --no-cpu-baseline --steps 60 --warmup 10` and its per-kernel summary (`tools/prof_summary.py`): the kernels of the timed region, one after the other |
| `{R}/bench_student_n1.json`, `bench_teacher_n1.json`, `bench_joint_n1.json` | BASELINE configs 2, 3 and the config-5 shard (`--workload student|teacher|joint`) |
| `{R}/kernel_stats_student.txt`, `kernel_stats_teacher.txt`, `kernel_stats_joint.txt`, `kernel_stats_senet50.txt`, `kernel_stats_b256.txt` | the same rocprofv3 per-kernel summary (serial mode, 20 timed steps) for configs 2, 3, the config-5 shard, the SE-ResNet50 distillation step and the batch-256 step |
| `{R}/bench_distill_senet50_n1.json` | the reference's default teacher (`run_distillation.m:82`): `--teacher senet50`, one face per pair |
| `{R}/bench_distill_senet50_b256_n1.json`, `bench_distill_b256_n1.json` | north_star's batch 256 on ONE GPU (`--per-gpu-batch 256`), SE-ResNet-50 / ResNet-50 teacher |
| `{R}/bench_distill_13frames_senet50_n1.json` | SURVEY 8f row 1: 13 face frames per pair through the SE-ResNet50 teacher, max-aggregated |
| `{R}/bench_cpu_teacher.json` | BASELINE config 1: `--workload cpu-teacher` (ResNet-50 forward + loss / classerror heads, batch 32, host cores only); re-taken in its own gpurun call at `c0d2e5a`+ after the oracle's default OpenMP team became min(physical cores, cgroup quota) for every use ({c1['value']} img/s on {c1['cpu_baseline']['cores']} threads; the collection's line, 256 threads under the 16-core quota: 10.0) |
| `{R}/bench_distill_capi_1rank.json` | `XM_DEBUG_DIST=1 bench.py --parserv rccl-capi`: the library's own communicator (xm_parserv_push / sync) with a 1-rank group; `rccl_ranks` = {capi.get('rccl_ranks')} |
| `{R}/bench_distill_torch_1rank.json`, `bench_distill_capi_late_init.json` | the same single-rank run through `torch.distributed` ({tor1['value']} pairs/s), and with the library's communicator created AFTER the networks (`XM_PS_LATE=1`: {late['value']} pairs/s -- the call-order trap of `include/xmodal.h`) |
| `{R}/bench_student_w400_n1.json`, `bench_distill_senet50_w400_n1.json` | round 5: the reference's REAL default shape -- `numSeconds = 4`, 512 x 400 spectrograms (`run_distillation.m:74`): student at batch 64 ({w400['value']} samples/s = {pct(w400['value'] * 22.33 / 1e3 / 157.3)} of peak at 22.33 GFLOP per sample; the line's own fraction field still used the W = 300 FLOPs, fixed after the collection) and the SE-ResNet50 distillation step ({dw400['value']} pairs/s) |
| `{R}/bench_distill_senet50_b256_serial_hint1.json` | round 5: north_star's batch on ONE stream with the host's `XM_EXEC_SINGLE_STREAM` hint (what a MATLAB / MEX host runs): {sh1['value']} pairs/s ({sh1['ms_per_step']} ms) against {se256['value']} two-stream |
| `{R}/bench_distill_gpus2_gloo0.json` | `XM_DEBUG_DIST=gloo0 python bench.py --gpus 2` with NO launcher around it: the command starts its two ranks itself (both on this box's one GPU, exchange over gloo: a functional run of the N > 1 path; `n_gpus` 2, `world` {g2.get('world')}, `rccl_ranks` {g2.get('rccl_ranks')} -- no RCCL communicator carried that exchange --, `control_group` {g2.get('control_group')}; the throughput means nothing) |
| `{R}/wgrad_patch_s2_bench.txt` | round 5 (own gpurun calls, commit `736abaa`): conv2's filter derivative, generic kernel against `conv_wgrad_patch_s2_kernel<5, 2>` (DESIGN.md 2.1h) at 32 / 64 / 256 spectrograms, and the steps with the old kernels / + this kernel / + the 3 x 3 patch kernel from 4096 output columns; one stream against two at batch 256 |
| `{R}/dgrad_s2_bench.txt` | round 5 (own gpurun call, commit `37160fb` before the launch-size rule): conv2's dgrad, merged stride-parity launch against `conv_dgrad_s2_kernel` forced at every batch (DESIGN.md 2.1i) |
| `{R}/stem3_bench.txt` | round 5 (own gpurun calls, AFTER the collection: commit `abf5347`): the teachers' conv1 through the implicit GEMM and through `conv_stem3_kernel` (DESIGN.md 2.1k) at 32 / 128 / 256 faces, the steps with and without it, the kernel without its stores, with a start offset between the two blocks of a CU |
| `{R}/dma_kernel_dissection.txt` | round 5 (own gpurun call, timing-only builds on `213741a`): per-layer table of the SE-ResNet50 at 256 faces, every tile configuration on the three 1 x 1 shapes, `conv_gemm_dma_kernel` without its A loads / B loads / epilogue / MFMAs, grid and ring variants; per-layer table of the student at 256 (DESIGN.md 2.1j) |
| `{R}/deferred_stores_and_halo64.txt` | round 5 (own gpurun call; built, parity-green, measured, not kept): the LDS-DMA kernel with its epilogue stores spread over the next tile, a 64-row halo-patch variant for the res2 3 x 3 layers, and the steps with both (DESIGN.md 2.1j) |
| `{R}/pmc_summary_senet50_b256.txt`, `pmc_traffic_senet50_b256.json` | round 5: the three PMC passes on north_star's batch: `conv_dgrad_s2_kernel` WRITE {tb['xm::conv_dgrad_s2_kernel<1>']['write_bytes']/1e6:.0f} MB per launch for 904 MB of dX (the merged launch: 2.1 x), `conv_wgrad_patch_s2_kernel` FETCH x 2 {tb['xm::conv_wgrad_patch_s2_kernel<5, 2>']['fetch_bytes_x2']/1e6:.0f} MB for 1489 MB of x + dY (the generic kernel: 4.2 x) |
| `{R}/kernel_stats_senet50_b256.txt` | rocprofv3 per-kernel summary of north_star's configuration (SE-ResNet50 teacher, 256 pairs, serial mode) |
| `{R}/pmc_summary.txt`, `{R}/pmc_traffic.json` | `rocprofv3 --pmc` passes (SQ counters; FETCH_SIZE; WRITE_SIZE — three separate runs, kernel trace only), per-launch averages per kernel (`tools/pmc_table.py`); `bench.py` reads `roofline.traffic` (+ the commit) from the JSON |

## Headline (default bench line)

* **{d['value']} pairs/s** on 1 MI355X ({d['ms_per_step']} ms/step over {d['steps']} timed steps = {d['steps'] * d['ms_per_step'] / 1e3:.2f} s;
  32 faces 224x224x3 + 32 spectrograms 512x300; fp32) = {d['model_tflops_per_gpu']} model-TFLOP/s =
  **{pct(d['model_frac_of_fp32_mfma_peak'])} of the 157.3 TFLOP/s fp32-MFMA peak for the WHOLE step** (teacher fwd + student fwd/bwd +
  BN/pool/loss/SGD; algorithmic 24.345 GFLOP per pair, conv + FC only).  Windows inside the timed region
  ({w.get('n')} x {w.get('steps_each')} steps): min {w.get('min')} / median {w.get('median')} / max {w.get('max')} pairs/s.
  With the driver's flags (`--steps 20 --warmup 5`): {dd['value']} pairs/s.  One stream: {ds['value']} pairs/s ({ds['ms_per_step']} ms).
* dominant kernel `{r['kernel']}`: **{r['achieved']} TFLOP/s = {pct(r['frac'])} of the fp32-MFMA roofline**
  ({r['launches']} launches, avg {r['avg_launch_ms']} ms, {r['flop_per_launch']/1e9:.2f} GFLOP algorithmic per launch; HIP events on the
  launch stream, serial pass); `kernel_stats.txt` (rocprofv3, same command) has it at {rocavg:.1f} us average.
  All convolution kernels together: {r['all_conv_kernels']['achieved']} TFLOP/s over {r['all_conv_kernels']['ms_per_step']} ms per step.
  HBM traffic of that kernel from the PMC passes (commit `{r.get('traffic_profile_commit')}`): {(r.get('traffic') or 0)/1e6:.0f} MB per launch.
* north_star's own configuration inside the same invocation (`north_star_b256`: SE-ResNet50 teacher + student, 256 pairs on
  this GPU, 12 timed steps): **{ns.get('value')} pairs/s = {pct(ns.get('model_frac') or 0)}** of the fp32-MFMA peak for the whole step.
* CPU baseline (oracle fp32 path = MatConvNet-CPU-equivalent restatement): {c['value']} pairs/s = {c.get('gflops')} GFLOP/s on {c['cores']} cores
  of {c.get('cpu')} ({c['sample']}).  Config 1 by itself: {c1['value']} images/s ({c1['cpu_baseline'].get('gflops')} GFLOP/s).
* `{tot}` (serial mode under rocprofv3).

| kernel (serial pass, HIP events) | ms / step | TFLOP/s | launches / step |
|---|---|---|---|
{rows}

| config | per-GPU batch | throughput | ms/step | whole-step fraction of fp32-MFMA peak |
|---|---|---|---|---|
| 4 (default): frozen ResNet50 teacher -> VGGVox student distillation step | 32 pairs | {d['value']} pairs/s | {d['ms_per_step']} | {pct(d['model_frac_of_fp32_mfma_peak'])} |
| 4, one stream (`--serial`) | 32 pairs | {ds['value']} pairs/s | {ds['ms_per_step']} | {pct(ds['model_frac_of_fp32_mfma_peak'])} |
| 4 with the reference's default teacher (SE-ResNet50), one face per pair | 32 pairs | {se['value']} pairs/s | {se['ms_per_step']} | {pct(se['model_frac_of_fp32_mfma_peak'])} |
| north_star batch 256 on one GPU, ResNet50 teacher | 256 pairs | {r256['value']} pairs/s | {r256['ms_per_step']} | {pct(r256['model_frac_of_fp32_mfma_peak'])} |
| north_star batch 256 on one GPU, SE-ResNet50 teacher | 256 pairs | {se256['value']} pairs/s | {se256['ms_per_step']} | {pct(se256['model_frac_of_fp32_mfma_peak'])} |
| 2: VGGVox student fwd+bwd+update | 64 | {st['value']} samples/s | {st['ms_per_step']} | {pct(st['model_frac_of_fp32_mfma_peak'])} |
| 3: SE-ResNet50 teacher fwd (2 sample-slice lanes) | 128 | {te['value']} img/s | {te['ms_per_step']} | {pct(te['model_frac_of_fp32_mfma_peak'])} |
| 5 shard: SE-ResNet50 fwd+bwd + student fwd+bwd | 64 pairs | {jo['value']} pairs/s | {jo['ms_per_step']} | {pct(jo['model_frac_of_fp32_mfma_peak'])} |
| 8f-1: 13 frames/pair, SE-ResNet50 teacher + student step | 32 pairs (416 faces) | {mf['value']} pairs/s | {mf['ms_per_step']} | {pct(mf['model_frac_of_fp32_mfma_peak'])} |
| 1: ResNet50 fwd + heads, batch 32, CPU restatement | 32 | {c1['value']} img/s | {c1['ms_per_step']} | n/a |

Round 4 -> round 5 (`r04/` -> `{R}/`, both builder-run on boxes of the same pool, which differ by 1-2 %): default line 3951 -> {d['value']} pairs/s;
north_star batch 256 (SE-ResNet50) 4257 -> {se256['value']} (`north_star_b256` on the default line: 4260 -> {ns.get('value')} = {pct(ns.get('model_frac') or 0)}); batch 256 with the
ResNet-50 teacher 4334 -> {r256['value']} ({pct(r256['model_frac_of_fp32_mfma_peak'])}); student batch 64: 5866 -> {st['value']}; config-5 shard 1851 -> {jo['value']}; config 3 12083 -> {te['value']} (no kernel of
that path changed).  Where it came from (DESIGN.md 2.1h, 2.1i): conv2's filter derivative from an LDS patch (`conv_wgrad_patch_s2_kernel`: 114 -> 125 TFLOP/s
at 256 spectrograms, 101 -> 113 at 32), conv2's dgrad with both row parities per wave and whole-line stores (`conv_dgrad_s2_kernel`: 115 -> 124 at 256; taken
from 6 rounds of blocks), the 3 x 3 patch filter derivative also for launches of >= 4096 output columns.  At batch 256 every kernel fills the chip: one
stream {sh1['ms_per_step']} ms against {se256['ms_per_step']} ms on two, and a kernel's own gain arrives 1 : 1; at 32 pairs the two-stream step is work-conserving and the side
stream's kernels are off the critical path ({ds['ms_per_step']} ms on one stream -> {d['ms_per_step']} ms overlapped), which is why the default line moved by less than the
boxes differ.  What was measured and NOT kept (DESIGN.md 2.1j): deferred epilogue stores in the LDS-DMA kernel, a 64-row halo variant, fewer
persistent DMA blocks per CU, a lower eight-wave threshold, register-double-buffered fragments in the patch kernel, a finer decomposition of the
dgrad kernel.  Added after the collection (commit `abf5347`, `stem3_bench.txt`): `conv_stem3_kernel` for the teachers' first layer (72-75 -> 82-84 TFLOP/s
at 256 faces; north_star step - 0.1 ms) and the tuning-table entries that select it.
`cpu_baseline`: the OpenMP team is now sized by min(affinity, cgroup CPU quota) -- {c['cores']} threads under a quota of {c.get('quota')} cores on this box:
{c['value']} pairs/s, min {c.get('min')} / max {c.get('max')} (round 4: 128 threads under the same quota, 1.5 ... 3.6 pairs/s from box to box).

## How the numbers were taken
```
python bench.py                                           # value, roofline (2nd, serial pass with HIP events), cpu_baseline
rocprofv3 --kernel-trace --stats -d ... -- python bench.py --serial --exec-hint 0 --no-cpu-baseline --steps 60 --warmup 10
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY \\
          SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA --output-format csv -- python bench.py --serial --steps 3 ...
rocprofv3 --kernel-trace --pmc FETCH_SIZE  ...            # separate pass
rocprofv3 --kernel-trace --pmc WRITE_SIZE  ...            # separate pass
```
Kernel durations (HIP events and rocprofv3 alike) are only meaningful when kernels do not share the chip,
so both are taken in `--serial` mode; the throughput `value` is taken with the overlap streams on.
`pmc_summary.txt`: `mfma%` = MFMA-pipe busy cycles / (4 SIMDs x CU-busy cycles); FETCH_SIZE is doubled per
MI355X_MICROARCH.md (gfx950 under-reports wide reads by 2x; upper estimate for the dword gathers).
"""
open(os.path.join(root, "profiles", "README.md"), "w").write(txt)
print("wrote profiles/README.md")
