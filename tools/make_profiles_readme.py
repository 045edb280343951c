#!/usr/bin/env python
"""Regenerates profiles/README.md from the JSON / text evidence under profiles/<round>/.
usage: python tools/make_profiles_readme.py r02"""
import json, os, re, sys
R = sys.argv[1] if len(sys.argv) > 1 else "r04"
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P = os.path.join(root, "profiles", R)
def L(n): return json.loads(open(os.path.join(P, n + ".json")).read().strip().splitlines()[-1])
d, dd, ds, st, te, jo, mf = (L("bench_distill_n1"), L("bench_distill_driver_flags_n1"), L("bench_distill_serial_n1"),
                             L("bench_student_n1"), L("bench_teacher_n1"), L("bench_joint_n1"),
                             L("bench_distill_13frames_senet50_n1"))
se, se256, r256, c1, capi = (L("bench_distill_senet50_n1"), L("bench_distill_senet50_b256_n1"), L("bench_distill_b256_n1"),
                             L("bench_cpu_teacher"), L("bench_distill_capi_1rank"))
tor1, late = L("bench_distill_torch_1rank"), L("bench_distill_capi_late_init")
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

Everything under `{R}/` (except the three files marked below) comes from ONE `gpurun` call on a fresh MI355X box at commit `{meta.get('commit')}`:
`bash tools/collect_profiles.sh {R} <commit>` (the script lists every command); this file is generated from
those files by `tools/make_profiles_readme.py {R}`.  `r01/` ... `r03/` are the previous rounds' evidence, unchanged.

| file | what |
|---|---|
| `{R}/pytest_gpu.txt` | `python -m pytest tests -q -m gpu` on that box ({tests}) |
| `{R}/bench_distill_n1.json` | `python bench.py` — default line: distillation step, 32 pairs/GPU, K chosen for a >= 2 s timed region, incl. `roofline`, `cpu_baseline`, `windows` |
| `{R}/bench_distill_driver_flags_n1.json` | `python bench.py --steps 20 --warmup 5` — the driver's command line (settle phase in front of the 20 timed steps: `settle_steps` in the line) |
| `{R}/bench_distill_serial_n1.json` | `python bench.py --serial` — same step on ONE HIP stream (no overlap), the mode the roofline leg and the profiles below use |
| `{R}/bench_under_rocprof.json`, `{R}/kernel_stats.txt` | `rocprofv3 --kernel-trace --stats -- python bench.py --serial --no-cpu-baseline --steps 60 --warmup 10` and its per-kernel summary (`tools/prof_summary.py`) |
| `{R}/bench_student_n1.json`, `bench_teacher_n1.json`, `bench_joint_n1.json` | BASELINE configs 2, 3 and the config-5 shard (`--workload student|teacher|joint`) |
| `{R}/kernel_stats_student.txt`, `kernel_stats_teacher.txt`, `kernel_stats_joint.txt`, `kernel_stats_senet50.txt`, `kernel_stats_b256.txt` | the same rocprofv3 per-kernel summary (serial mode, 20 timed steps) for configs 2, 3, the config-5 shard, the SE-ResNet50 distillation step and the batch-256 step |
| `{R}/bench_distill_senet50_n1.json` | the reference's default teacher (`run_distillation.m:82`): `--teacher senet50`, one face per pair |
| `{R}/bench_distill_senet50_b256_n1.json`, `bench_distill_b256_n1.json` | north_star's batch 256 on ONE GPU (`--per-gpu-batch 256`), SE-ResNet-50 / ResNet-50 teacher |
| `{R}/bench_distill_13frames_senet50_n1.json` | SURVEY 8f row 1: 13 face frames per pair through the SE-ResNet50 teacher, max-aggregated |
| `{R}/bench_cpu_teacher.json` | BASELINE config 1: `--workload cpu-teacher` (ResNet-50 forward + loss / classerror heads, batch 32, host cores only) |
| `{R}/bench_distill_capi_1rank.json` | `XM_DEBUG_DIST=1 bench.py --parserv rccl-capi`: the library's own communicator (xm_parserv_push / sync) with a 1-rank group; `rccl_ranks` = {capi.get('rccl_ranks')} |
| `{R}/bench_distill_torch_1rank.json`, `bench_distill_capi_late_init.json` | the same single-rank run through `torch.distributed` ({tor1['value']} pairs/s), and with the library's communicator created AFTER the networks (`XM_PS_LATE=1`: {late['value']} pairs/s -- the call-order trap of `include/xmodal.h`) |
| `{R}/phase_marks.txt` | `XM_BENCH_MARKS=1`: timing events on the main stream at the phase boundaries of the student step (forward / backward / join / update / gap to the next step) for the default line and its one-stream / no-side-stream / no-teacher-overlap variants |
| `{R}/stem_bench.txt` | the student's conv1 through `conv_stem_kernel` and through the implicit-GEMM kernel, with / without batch moments; `tools/store_mfma_probe.hip`: stores / MFMAs / both for the same tile shape, three output layouts, two tile orders (DESIGN.md 2.1e) |
| `{R}/schedule_experiments.txt` | A/B lines behind DESIGN.md 2.3b: teacher pass over 64 / 128 / 256 faces, `--teacher-gate`, wgrad deferred behind dgrad, stream priorities, and every round-3 kernel change switched off by its environment variable |
| `{R}/stem_bwd_bench.txt` | round 4: `tools/stem_bwd_bench.py` (conv1's backward chain at 32 / 64 / 256 spectrograms: bnorm + relu + pool backward and the filter derivative as two passes against `xm_nnconv_backward_filter_bnrelupool`, DESIGN.md 2.2e) and `tools/mall_chunk_bench.py` (the same pair in sample chunks that would fit the Infinity Cache: no gain) |
| `{R}/bench_distill_gpus2_gloo0.json` | round 4: `XM_DEBUG_DIST=gloo0 python bench.py --gpus 2` with NO launcher around it: the command starts its two ranks itself (both on this box's one GPU, exchange over gloo: a functional run of the N > 1 path, `n_gpus` 2 / `rccl_ranks` 2 in the line; the throughput means nothing) |
| `{R}/w8_bench.txt` | round 4: `tools/conv_bench.py --cfg 0 / --cfg 7`: the 128 x 128 tile by four waves of 222 VGPRs against eight waves of 128 (DESIGN.md 2.1f), forward and dgrad, 32 / 64 / 256 samples, idle device |
| `{R}/w8_threshold_tables.txt` | round 4, ANOTHER gpurun call (another box, before the main collection): the bench lines with a tuning table generated for each `XM_W8_MIN_TILES` setting (from 1024 tiles / always / never), DESIGN.md 2.1f |
| `{R}/wgrad_patch_bench.txt` | round 4: filter derivative of the student's 3 x 3 layers, generic kernel (`XM_NO_WGRAD_PATCH=1`) against `conv_wgrad_patch_kernel<30>` (DESIGN.md 2.1g), 32 / 64 / 256 spectrograms, idle device |
| `{R}/timeline_student64_generic_wgrad.csv`, `timeline_student64_patch_wgrad_on_side_stream.csv` | round 4: `rocprofv3 --kernel-trace` of `bench.py --workload student` (last 40 ms: start ns, end ns, queue, stream, kernel) with the generic filter-derivative kernel and with the patch kernel forced onto the side stream: what runs next to what (DESIGN.md 2.1g; taken at `0da31f7` + the kernel, before the one-stream rule) |
| `{R}/halo_bench.txt`, `stats_bench.txt`, `bnbwd_bench.txt` | `tools/halo_bench.py` (halo-patch variants vs the best implicit-GEMM configuration, 32 / 64 / 256 samples), `tools/stats_bench.py` (conv with / without fused batch moments), `tools/bnbwd_bench.py` (bnorm backward chains) |
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

Round 3 -> round 4 on the default line: 3924 -> {d['value']} pairs/s; student batch 64: 5818 -> {st['value']} samples/s; north_star batch 256
(SE-ResNet50): 4197 -> {se256['value']} pairs/s; config-5 shard: 1750 -> {jo['value']} pairs/s; one stream: 3574 -> {ds['value']}.  Where it came from
(DESIGN.md 2.2e, 2.2f, 2.1f, 2.1g): conv1's filter derivative computed straight through bnorm + relu + pool (`conv_stem_wgrad_bnp_kernel`: the
462 MB derivative of the student's first layer is neither written nor read: chain 0.49 -> 0.38 ms at 32 spectrograms, 3.6 -> 2.8 ms at 256,
`stem_bwd_bench.txt`); the SE tail of a TRAINED teacher fused in both directions (backward 13 -> 8 passes over the block's tensors, forward
6 -> 4: every line of config 5); the 128 x 128 tile by eight waves of 128 VGPRs for launches of >= 1024 tiles (`w8_bench.txt`: +3 ... 4 % alone);
the filter derivative of the student's 3 x 3 layers from an input patch for one-stream callers (`wgrad_patch_bench.txt`: 104 -> 126 TFLOP/s
at 256 spectrograms; the one-stream line above).  `schedule_experiments.txt` has every switch of this and the previous round as an A/B
line: the eight-wave configuration never / from 1024 tiles / always, the patch kernel on one and on two streams, the halo selection margin
at 4 / 1.5 / 0 %, the frozen teacher in slices.  What the A/Bs and the two timelines say (DESIGN.md 2.1f, 2.1g): MFMA-bound kernels side by side
take the sum of their times, so inside the two-stream step a kernel that is faster alone is worth nothing unless it removes work from the main
stream; the step is work-conserving (one stream {ds['ms_per_step']} ms -> {d['ms_per_step']} ms overlapped) and its main stream never waits (`phase_marks.txt`).

## How the numbers were taken
```
python bench.py                                           # value, roofline (2nd, serial pass with HIP events), cpu_baseline
rocprofv3 --kernel-trace --stats -d ... -- python bench.py --serial --no-cpu-baseline --steps 60 --warmup 10
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
