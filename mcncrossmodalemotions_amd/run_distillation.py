"""run_distillation mirror (emoVoxCeleb/run_distillation.m).

    [net, info] = run_distillation('gpus', 2, 'numSeconds', 4, 'batchSize', 64, ...)

Same option names, defaults and flow as the reference entry point: build the imdb, name the
experiment directory, build the student with emoVoxZoo, bind getBatchEmoVoxCeleb, call cnn_train_dag
with the reference's arguments (run_distillation.m:170-182).  What differs, because there is no
VoxCeleb / MatConvNet here: the imdb is the seeded synthetic stand-in of batch.SyntheticEmoVoxImdb
(teacher logits per track, wav lengths), spectrogram magnitudes come from a seeded generator, and
`gpus` is the torchrun world (one process per GPU).  Extensions are keyword-only and marked.
"""
import os

import numpy as np

from . import batch as xbatch
from . import train, zoo


def run_distillation(gpus=(2,), cont=True, miniVal=0.2, numSeconds=4, batchSize=64, numEpochs=300,
                     miniEpochRatio=None, numPredEmotions=8, fromScratch=True, logitAggregator="max",
                     datasetName="voxceleb", teacher="senet50-ferplus", student="emovoxceleb-student",
                     lossType="hot-cross-ent", temperature=2, fixedSegments=False, learningRate=None,
                     parameterServer="tmove", wavDir=None,
                     *, imdb=None, dataDir="data/xEmo18", numTracks=256, widthMult=1.0, seed=0, verbose=False):
    """Options as in run_distillation.m:72-90.  Extensions (keyword-only): `imdb` (a prepared imdb),
    `dataDir` (root of the experiment directories), `numTracks` / `seed` (synthetic imdb), `widthMult`
    (narrow student for tests)."""
    gpus = list(np.atleast_1d(gpus))
    if miniEpochRatio is None:
        miniEpochRatio = 0.05 * len(gpus)                      # :77
    if learningRate is None:
        learningRate = np.logspace(-4, -5, numEpochs)          # :87
    if imdb is None:
        imdb = xbatch.SyntheticEmoVoxImdb(num_tracks=numTracks, seed=seed, num_emotions=8,
                                          min_seconds=numSeconds + 0.5, max_seconds=numSeconds + 5.0,
                                          val_fraction=0.25)
    # experiment directory (:93-104) -- opts.temperature only names it (emoVoxZoo.m:152 hard-codes T = 2)
    sname = "%s-%s" % (student, lossType) + ("-scratch" if fromScratch else "")
    expName = "voxceleb-%s-%s-%dsec-%demo-agg-%s" % (teacher, sname, numSeconds, numPredEmotions, logitAggregator)
    expDir = os.path.join(dataDir, expName)
    if lossType == "hot-cross-ent":
        expDir += "-temp%d" % temperature
    # ParameterServer before the first device allocation / kernel of the process (xmodal.h "CALL ORDER": a communicator
    # created later slows every step); 'tmove' (run_distillation.m:88) = the library's communicator when there is
    # more than one worker and RCCL can carry it -- every worker of the node on its own device --, torch.distributed
    # otherwise (gloo test groups on CPU / on one shared GPU).  ONE RCCL communicator per worker: a multi-GPU host
    # initialises torch.distributed on GLOO (it is only the control plane: store, barriers) and the library's
    # communicator carries the exchange; an nccl process group is accepted but is a second communicator.
    import torch
    import torch.distributed as dist
    if parameterServer == "tmove":
        # decided once for all workers from the (host, device) pairs they post to the store (train.ParameterServer.
        # choose_backend): the library's communicator when every worker has its own device, torch.distributed otherwise
        parameterServer = train.ParameterServer.choose_backend()
    if isinstance(parameterServer, train.ParameterServer):
        parserv = parameterServer
        parserv.start()
    else:   # all workers agree on the backend (a failed communicator start falls back to torch.distributed everywhere)
        parserv = train.ParameterServer.start_agreed(parameterServer)
    net = zoo.emoVoxZoo(student, scratch=1 if fromScratch else 0, lossType=lossType, numSeconds=numSeconds,
                        numOutputs=numPredEmotions, width_mult=widthMult)             # :125-129
    net.meta.setdefault("augmentation", {})["transformation"] = "I"                  # :130
    trainSamples = [i for i in range(len(imdb.set)) if imdb.set[i] == 1]              # :137-138
    valSamples = [i for i in range(len(imdb.set)) if imdb.set[i] == 2]
    if miniVal < 1 and valSamples:                                                    # :141-146
        rng = np.random.default_rng(0)
        pick = rng.choice(len(valSamples), int(round(len(valSamples) * miniVal)), replace=False)
        valSamples = [valSamples[i] for i in sorted(pick)]
    epochSize = len(trainSamples) * miniEpochRatio                                    # :154
    brng = np.random.default_rng(seed + 17)

    def getBatch(imdb_, batch):                                                       # getBatchFn, :210-224
        return xbatch.getBatchEmoVoxCeleb(imdb_, batch, imageSize=(512, int(round(numSeconds * 100))),
                                          numPredEmotions=numPredEmotions, logitAggregator=logitAggregator,
                                          lossType=lossType, transformation=net.meta["augmentation"]["transformation"],
                                          rng=brng, fixedSegments=fixedSegments)   # bopts.fixedSegments, :220
        # (fixedSegments = true fails upstream as well: getBatchEmoVoxCeleb.m:15 passes timeOffsets = [])

    return train.cnn_train_dag(net, imdb, getBatch, learningRate=learningRate, batchSize=batchSize,
                               numEpochs=numEpochs, train=trainSamples, val=valSamples, cont=cont,
                               expDir=expDir, epochSize=epochSize,
                               parameterServer=parserv,
                               extractStatsFn=train.extractStats, verbose=verbose)
