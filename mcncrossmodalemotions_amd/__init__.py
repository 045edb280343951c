"""mcncrossmodalemotions_amd -- MI355X-native (gfx950) hot path of albanie/mcnCrossModalEmotions.

Layout:
    csrc/      hand-written HIP kernels + the C ABI (include/xmodal.h) -> libxmodal_hip.so
    _lib.py    ctypes binding (no fallback: fails loudly when the library is missing)
    vl.py      host-side mirror of the MatConvNet / mcnExtraLayers operator API (vl_nn*)
    dagnn.py   host-side mirror of dagnn.DagNN / dagnn.Layer (eval, forward/backward)
    zoo.py     emoVoxZoo / ferPlusZoo mirrors (student and teacher graphs)
    batch.py   getBatchEmoVoxCeleb / getImageBatch mirrors on synthetic data
    train.py   cnn_train_dag step mirror (SGD + ParameterServer -> RCCL all-reduce)
"""
__version__ = "0.1.0"
