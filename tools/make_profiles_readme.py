#!/usr/bin/env python
"""Regenerates profiles/README.md from the JSON / text evidence under profiles/<round>/.
usage: python tools/make_profiles_readme.py r02"""
import json, os, re, sys
R = sys.argv[1] if len(sys.argv) > 1 else "r06"
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P = os.path.join(root, "profiles", R)
def L(n): return json.loads(open(os.path.join(P, n + ".json")).read().strip().splitlines()[-1])
d, dd, ds, st, te, jo, mf = (L("bench_distill_n1"), L("bench_distill_driver_flags_n1"), L("bench_distill_serial_n1"),
                             L("bench_student_n1"), L("bench_teacher_n1"), L("bench_joint_n1"),
                             L("bench_distill_13frames_senet50_n1"))
se, se256, r256, c1, capi = (L("bench_distill_senet50_n1"), L("bench_distill_senet50_b256_n1"), L("bench_distill_b256_n1"),
                             L("bench_cpu_teacher"), L("bench_distill_capi_1rank"))
tor1, late = L("bench_distill_torch_1rank"), L("bench_distill_capi_late_init")
imdb_row = ""
if os.path.exists(os.path.join(P, "bench_distill_imdb_windows_senet50_n1.json")):
    iw = L("bench_distill_imdb_windows_senet50_n1")
    imdb_row = ("| `%s/bench_distill_imdb_windows_senet50_n1.json` | the same with RAGGED windows as getBatch draws them from an imdb "
                "(`--imdb-windows 1`: `time2idx` rows of a random 3 s window of a 4-20 s track, clamped to the rows the track has): "
                "%s pairs/s, %s |\n" % (R, iw["value"], iw["config"]["workload"].split(", imdb windows: ")[-1]))
w400, dw400, sh1, g2 = (L("bench_student_w400_n1"), L("bench_distill_senet50_w400_n1"), L("bench_distill_senet50_b256_serial_hint1"),
                        L("bench_distill_gpus2_gloo0"))
tb = json.load(open(os.path.join(P, "pmc_traffic_senet50_b256.json")))
o256, o32 = L("bench_distill_senet50_b256_round5_stem"), L("bench_distill_round5_stem")
chain = open(os.path.join(P, "stem_chain_bench.txt")).read().strip().splitlines()
ks256 = open(os.path.join(P, "kernel_stats_senet50_b256.txt")).read()
def kus(name):
    m_ = re.search(re.escape(name) + r".*?\s(\d+)\s+([\d.]+)\s+([\d.]+)\s+[\d.]+\n", ks256)
    return float(m_.group(3)) if m_ else float("nan")
ns = d.get("north_star_b256") or {}
r, c = d["roofline"], d["cpu_baseline"]
tests = [l for l in open(os.path.join(P, "pytest_gpu.txt")).read().strip().splitlines() if "passed" in l][-1]
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
those files by `tools/make_profiles_readme.py {R}`.  `r01/` ... `r05/` are the previous rounds' evidence, unchanged (the schedule /
priority / variant A/Bs of rounds 3-5 -- `r03/schedule_experiments.txt`, `r04/w8_bench.txt`, `r05/dma_kernel_dissection.txt`,
`r05/wgrad_patch_s2_bench.txt` ... -- were not repeated: no kernel they cover changed).  The boxes of the pool differ by up to 4 % from call to
call (the round-5 kernels, run as the other arm IN THE SAME CALL, give {o256['value']} pairs/s on north_star's line here, 4115 on the box of this
round's first collection at `e4e1a38`, 4344 in round 5's collection); every comparison below is same-call.

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
{imdb_row}| `{R}/bench_cpu_teacher.json` | BASELINE config 1: `--workload cpu-teacher` (ResNet-50 forward + loss / classerror heads, batch 32, host cores only): {c1['value']} img/s on {c1['cpu_baseline']['cores']} threads |
| `{R}/bench_distill_capi_1rank.json` | `XM_DEBUG_DIST=1 bench.py --parserv rccl-capi`: the library's own communicator (xm_parserv_push / sync) with a 1-rank group; `rccl_ranks` = {capi.get('rccl_ranks')} |
| `{R}/bench_distill_torch_1rank.json`, `bench_distill_capi_late_init.json` | the same single-rank run through `torch.distributed` ({tor1['value']} pairs/s), and with the library's communicator created AFTER the networks (`XM_PS_LATE=1`: {late['value']} pairs/s -- the call-order trap of `include/xmodal.h`) |
| `{R}/bench_student_w400_n1.json`, `bench_distill_senet50_w400_n1.json` | the reference's REAL default shape -- `numSeconds = 4`, 512 x 400 spectrograms (`run_distillation.m:74`): student at batch 64 ({w400['value']} samples/s = {pct(w400['model_frac_of_fp32_mfma_peak'])} of peak at 22.33 GFLOP per sample: re-taken with the fixed FLOP count) and the SE-ResNet50 distillation step ({dw400['value']} pairs/s) |
| `{R}/bench_distill_senet50_b256_serial_hint1.json` | north_star's batch on ONE stream with the host's `XM_EXEC_SINGLE_STREAM` hint (what a MATLAB / MEX host runs): {sh1['value']} pairs/s ({sh1['ms_per_step']} ms) against {se256['value']} two-stream |
| `{R}/bench_distill_gpus2_gloo0.json` | `XM_DEBUG_DIST=gloo0 python bench.py --gpus 2` with NO launcher around it (both ranks on this box's one GPU, exchange over gloo: a functional run of the N > 1 path; `n_gpus` 2, `world` {g2.get('world')}, `rccl_ranks` {g2.get('rccl_ranks')}, `control_group` {g2.get('control_group')}; the 8-rank run is `tests/test_bench_launch.py::test_gpus_8_functional_run_on_one_gpu`) |
| `{R}/stem_chain_bench.txt` | round 6 (DESIGN.md 2.4): the student's conv1 -> bn1 -> relu1 -> pool1 chain at 32 / 64 / 256 spectrograms, composed operators (round-5 kernels) against the Gram route, forward and backward, per launch on an idle device |
| `{R}/bench_distill_senet50_b256_round5_stem.json`, `bench_distill_round5_stem.json` | the other arm in the same call (`XM_NO_STEM_FWD=1 XM_NO_STEM_GRAM=1`: the round-5 stem kernels): north_star line {o256['value']} pairs/s ({o256['ms_per_step']} ms) against {se256['value']} ({se256['ms_per_step']} ms); default line {o32['value']} ({o32['ms_per_step']} ms) against {d['value']} ({d['ms_per_step']} ms) |
| `{R}/stem_chain_dissection.txt` | round 6 (own gpurun calls on the way): the six versions of `conv_stem_wgrad_pool_kernel` and the forward kernel with parts switched off -- what bounded each version (DESIGN.md 2.4) |
| `{R}/pmc_summary_senet50_b256.txt`, `pmc_traffic_senet50_b256.json` | the three PMC passes on north_star's batch: `conv_stem_bnpool_fwd_kernel` WRITE {tb['xm::conv_stem_bnpool_fwd_kernel']['write_bytes']/1e6:.0f} MB per launch for 1151 MB of pooled output + table, `conv_stem_wgrad_pool_kernel` FETCH x 2 {tb['xm::conv_stem_wgrad_pool_kernel<2, false>']['fetch_bytes_x2']/1e6:.0f} MB for 1151 MB of pooled derivative + table + 157 MB of input (the round-5 chain moved 15.2 GB) |
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

Round 5 -> round 6, same call (the round-5 stem kernels are the `XM_NO_STEM_FWD=1 XM_NO_STEM_GRAM=1` arm): north_star's line {o256['value']} -> **{se256['value']} pairs/s
({o256['ms_per_step']} -> {se256['ms_per_step']} ms, {pct(se256['value'] / o256['value'] - 1)} more)**, default line {o32['value']} -> {d['value']} ({pct(d['value'] / o32['value'] - 1)} more).  `stem_step_ab.txt` has the same pair on the other boxes the round touched.  Where it comes from
(DESIGN.md 2.4): conv1's output -- 3.7 GB at 256 spectrograms -- is no longer a tensor.  In the serial pass (`kernel_stats_senet50_b256.txt`):
`stem_gram_kernel` {kus('stem_gram_kernel<2>'):.0f} us + `conv_stem_bnpool_fwd_kernel` {kus('conv_stem_bnpool_fwd_kernel'):.0f} us + `conv_stem_wgrad_pool_kernel` {kus('conv_stem_wgrad_pool_kernel<2, false>'):.0f} us (+ 40 us of reductions) where round 5 had
`conv_stem_kernel` 1489 + `pool_fwd_lds_kernel` ~1000 + `bnpool_bwd_partial_pooled_kernel` ~500 + `conv_stem_wgrad_bnp_kernel` 2125 us.
Per launch on an idle device (`stem_chain_bench.txt`):
```
{chr(10).join(l for l in chain if l.startswith('N='))}
```
Student alone at batch 64: 5938 -> {st['value']} samples/s ({pct(st['model_frac_of_fp32_mfma_peak'])}); width 400: 4215 -> {w400['value']}; config-5 shard {jo['value']} pairs/s ({pct(jo['model_frac_of_fp32_mfma_peak'])}; only its student half
changed); config 3 {te['value']} img/s (no kernel of that path changed).
`cpu_baseline`: the OpenMP team is sized by min(affinity, cgroup CPU quota) -- {c['cores']} threads under a quota of {c.get('quota')} cores on this box:
{c['value']} pairs/s, min {c.get('min')} / max {c.get('max')}.

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
