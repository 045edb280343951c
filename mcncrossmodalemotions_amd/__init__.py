"""mcncrossmodalemotions_amd -- MI355X-native (gfx950) hot path of albanie/mcnCrossModalEmotions.

Layout:
    csrc/      hand-written HIP kernels + the C ABI (include/xmodal.h) -> libxmodal_hip.so
    _lib.py    ctypes binding (no fallback: fails loudly when the library is missing)
    vl.py      host-side mirror of the MatConvNet / mcnExtraLayers operator API (vl_nn*)
    dagnn.py   host-side mirror of dagnn.DagNN / dagnn.Layer (eval, forward/backward)
    zoo.py     emoVoxZoo / ferPlusZoo mirrors (student and teacher graphs)
    batch.py   getBatchEmoVoxCeleb / getImageBatch mirrors on synthetic data
    train.py   cnn_train_dag mirror (SGD + ParameterServer -> RCCL all-reduce, bucketed + overlapped)
    run_distillation.py / external.py   the reference's entry points (driver, feature extraction)
"""
import os as _os

# The step runs on 3-4 HIP streams (main, wgrad side stream, teacher stream, RCCL's own).  ROCm maps
# streams onto GPU_MAX_HW_QUEUES hardware queues (default 4); once RCCL has taken its queues the compute
# streams start sharing one and the overlap is silently serialised (measured: 3380 -> 3080 pairs/s as soon
# as a process group exists).  Must be set before the HIP runtime initialises.  16 since round 2: with a process
# group AND the library's own communicator (--parserv rccl-capi) 8 queues serialise again (2600 vs 3140 pairs/s),
# and the torch path with a process group gains too (3324 -> 3503 pairs/s).
_os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
_os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

__version__ = "0.1.0"
