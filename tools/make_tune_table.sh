#!/bin/bash
# Regenerates mcncrossmodalemotions_amd/tune_gfx950.txt: run through gpurun, then copy gpurun_out/tune_gfx950.txt
# over the tracked file.  Every workload of bench.py, the inference buckets and the GPU tests leave their shapes.
cd "$GRAFT_REPO_ROOT" || exit 1
export XM_TUNE_FILE=$PWD/gpurun_out/tune_gfx950.txt XM_TUNE_SAVE=1 XM_TUNE_REPS=8   # best of 8 launches per configuration
rm -f $XM_TUNE_FILE
A="--no-cpu-baseline --no-roofline --steps 20 --warmup 3"
python bench.py $A > /dev/null
python bench.py --serial $A > /dev/null
python bench.py --workload student $A > /dev/null
python bench.py --workload student --wgrad-stream 0 $A > /dev/null      # one-stream callers: conv_wgrad_patch_kernel's entries
python bench.py --workload teacher $A > /dev/null
python bench.py --workload teacher --serial $A > /dev/null
python bench.py --workload joint $A > /dev/null
python bench.py --teacher senet50 $A > /dev/null
python bench.py --teacher senet50 --serial $A > /dev/null
python bench.py --per-gpu-batch 256 --steps 5 --warmup 2 --no-cpu-baseline --no-roofline > /dev/null
python bench.py --serial --per-gpu-batch 256 --steps 5 --warmup 2 --no-cpu-baseline --no-roofline > /dev/null
python bench.py --teacher senet50 --per-gpu-batch 256 --steps 5 --warmup 2 --no-cpu-baseline --no-roofline > /dev/null
python bench.py --frames 13 --teacher senet50 --steps 5 --warmup 2 --no-cpu-baseline --no-roofline > /dev/null
# the reference's real default shape: numSeconds = 4 -> 512 x 400 spectrograms (run_distillation.m:74), batch 64 and the 32-pair shard
python bench.py --workload student --width 400 $A > /dev/null
python bench.py --workload student --width 400 --serial $A > /dev/null
python bench.py --width 400 $A > /dev/null
python bench.py --teacher senet50 --width 400 $A > /dev/null
python - <<PY
import numpy as np, torch
from mcncrossmodalemotions_amd import vl, zoo, external
rng = np.random.default_rng(0)
net = zoo.emoVoxZoo(numSeconds=3, scratch=0)
net.move("gpu"); net.mode = "test"
specs = [vl.from_numpy(np.abs(rng.standard_normal((512, w))).astype(np.float32)) for w in range(100, 1101, 100)]
external.compute_audio_feats(net, specs)
external.compute_audio_feats(net, specs * 4, batch_by_bucket=True)
print(vl.tune_save())
PY
wc -l $XM_TUNE_FILE
