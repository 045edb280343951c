"""The numerical / structural pins the reference itself holds for the hot path.
There are no reference tests or golden vectors; these are the closed forms and tables in the
MATLAB sources, checked against the oracle AND the product's host-side mirrors."""
import numpy as np

from mcncrossmodalemotions_amd import batch, dagnn, zoo
from oracle import oracle as O


def test_pool6_bucket_table():
    """emoVoxZoo.m:258-259: buckets.pool = [2 5 8 11 14 17 20 23 27 30] for widths 100..1000.
    The student's conv/pool geometry must yield exactly these fc6 output widths."""
    assert zoo.BUCKETS_POOL == [2, 5, 8, 11, 14, 17, 20, 23, 27, 30]
    assert zoo.BUCKETS_WIDTH == list(range(100, 1001, 100))
    net = zoo.vggvox(width_mult=0.0625)
    for width, p1 in zip(zoo.BUCKETS_WIDTH, zoo.BUCKETS_POOL):
        h, w = 512, width
        for l in net.layers:
            b = l.block
            if isinstance(b, dagnn.Conv) and l.name != "fc7" and l.name != "fc8":
                fh, fw = b.size[0], b.size[1]
                pad = b.pad
                h = O.conv_out_size(h, pad[0], pad[1], fh, 1, b.stride[0])
                w = O.conv_out_size(w, pad[2], pad[3], fw, 1, b.stride[1])
            elif isinstance(b, dagnn.Pooling) and l.name != "pool6":
                h = O.conv_out_size(h, 0, 0, b.poolSize[0], 1, b.stride[0])
                w = O.conv_out_size(w, 0, 0, b.poolSize[1], 1, b.stride[1])
            if l.name == "fc6":
                break
        assert (h, w) == (1, p1), (width, h, w)
    # updatePooling (emoVoxZoo.m:256-269) sets pool6 from the same table
    n3 = zoo.emoVoxZoo(numSeconds=3, width_mult=0.0625)
    assert n3.getLayer("pool6").block.poolSize == [1, 8]
    n4 = zoo.emoVoxZoo(numSeconds=4, width_mult=0.0625)
    assert n4.getLayer("pool6").block.poolSize == [1, 11]


def test_time2idx_closed_form():
    """getBatchEmoVoxCeleb.m:210-214: idx = floor(max(25 t - 1, 0) / 6) + 1."""
    for t, ref in [(0.0, 1), (0.04, 1), (0.28, 2), (1.0, 5), (3.0, 13), (4.0, 17)]:
        assert O.time2idx(t) == ref and batch.time2idx(t) == ref


def test_audio_sample_count():
    """getBatchEmoVoxCeleb.m:67-68: (0.01 W + 0.025 - 0.001) * 16000."""
    assert round(O.aud_samples(300)) == 48384 and round(batch.aud_samples(300)) == 48384
    assert round(O.aud_samples(400)) == 64384 and round(batch.aud_samples(400)) == 64384


def test_class_order_and_loss_head():
    """emoVoxZoo.m:180-181 (class order), :152 (T = 2, logit targets), :160-169 (metric layers)."""
    assert zoo.EMOTIONS == ["neutral", "happiness", "surprise", "sadness", "anger", "disgust", "fear", "contempt"]
    net = zoo.emoVoxZoo(numSeconds=3, width_mult=0.0625)
    loss = net.getLayer("loss")
    assert isinstance(loss.block, dagnn.SoftmaxCELoss)
    assert loss.block.temperature == 2 and loss.block.logitTargets is True
    assert loss.inputs == ["prediction", "logitTarget"] and loss.outputs == ["objective"]
    assert net.getLayer("classerror").inputs == ["prediction", "maxLabel"]
    assert isinstance(net.getLayer("classAccs").block, dagnn.ErrorStats)
    assert sorted(net.getInputs()) == ["data", "logitTarget", "maxLabel"]
    assert net.meta["classes"]["name"] == zoo.EMOTIONS
    # the last FC has 8 outputs (emoVoxZoo.m:213-220)
    assert net.getLayer("fc8").block.size[3] == 8


def test_student_and_teacher_sizes():
    """SURVEY Appendix B parameter counts (16.62 M conv/FC student, 23.5 M / 26.0 M teachers)."""
    s = zoo.emoVoxZoo(numSeconds=3)
    conv = sum(int(np.prod(p.value.shape)) for k, p in s.params.items() if k.endswith("f"))
    assert conv == 4704 + 614400 + 884736 + 884736 + 589824 + 9437184 + 4194304 + 8192
    t = zoo.ferPlusZoo("resnet50-ferplus")
    assert t.getInputs()[0] == "data"  # ferPlusZoo.m:127-133
    n = sum(int(np.prod(p.value.shape)) for p in t.params.values())
    assert 23.4e6 < n < 23.7e6
    se = zoo.ferPlusZoo("senet50-ferplus")
    n = sum(int(np.prod(p.value.shape)) for p in se.params.values())
    assert 25.9e6 < n < 26.2e6
    zoo.strip_losses(t)
    assert t.getOutputs() == ["prediction"]  # fetch_emovoxceleb_imdb.m:101-106,130


def test_student_macs_match_survey():
    """SURVEY 8d work model: student forward = 2.8311 GMAC per 512x300 sample."""
    net = zoo.emoVoxZoo(numSeconds=3)
    h, w, macs = 512, 300, 0
    for l in net.layers:
        b = l.block
        if isinstance(b, dagnn.Conv):
            fh, fw, fc, k = b.size
            h = O.conv_out_size(h, b.pad[0], b.pad[1], fh, 1, b.stride[0])
            w = O.conv_out_size(w, b.pad[2], b.pad[3], fw, 1, b.stride[1])
            macs += h * w * k * fh * fw * fc
        elif isinstance(b, dagnn.Pooling):
            h = O.conv_out_size(h, 0, 0, b.poolSize[0], 1, b.stride[0])
            w = O.conv_out_size(w, 0, 0, b.poolSize[1], 1, b.stride[1])
    assert abs(macs - 2.8311e9) / 2.8311e9 < 1e-3


def test_crop_window_is_one_based():
    """getBatchEmoVoxCeleb.m:109-119,141-152: wr = randi(wd) lies in [1, wd] and enters starttime = wr / fs as a
    1-BASED sample number; a short clip uses wr = 1; the window of logit rows is time2idx(start) .. time2idx(end),
    clipped to the rows that exist.  Hand-computed boundary: time2idx steps from 1 to 2 at t = 0.28 s, i.e. at
    sample 4480 -- a 0-based draw (wr = 4479) would still map to row 1."""
    fs, aud = 16000, batch.aud_samples(300)

    class Fixed:                       # stands in for randi: returns the requested draw
        def __init__(self, v):
            self.v = v

        def integers(self, lo, hi):
            assert lo == 1, "randi(wd) draws from 1"
            self.hi = hi
            return self.v
    total = 8 * fs
    r = Fixed(4480)
    wr, s, e = batch.crop_window(total, aud, fs, 40, r)
    assert r.hi == total - 48384 + 1                     # numpy's exclusive upper bound: draws reach wd
    assert (wr, s) == (4480, 2)                          # 4480 / 16000 = 0.28 s -> floor((7 - 1) / 6) + 1 = 2
    assert e == batch.time2idx((4480 + 48384 - 1) / fs) == 14
    wr, s, e = batch.crop_window(total, aud, fs, 40, Fixed(4479))
    assert (wr, s) == (4479, 1)
    # endIdx is clipped to the cached rows (:152)
    assert batch.crop_window(total, aud, fs, 9, Fixed(4480))[2] == 9
    # short clip: wr = 1, never a draw (:113-118)
    class NoDraw:
        def integers(self, lo, hi):
            raise AssertionError("no random draw for a short clip")
    assert batch.crop_window(2 * fs, aud, fs, 40, NoDraw()) == (1, 1, batch.time2idx((1 + 48384 - 1) / fs))
    # clips are thresholded at DATASET_LIMIT = 19.9 s before the draw (:81-89)
    r = Fixed(1)
    batch.crop_window(30 * fs, aud, fs, 200, r)
    assert r.hi == int(19.9 * fs) - 48384 + 1
    # fixedSegments (:91-101,136-137): wr = offset * fs + 1, all rows; upstream's empty timeOffsets is an index error
    assert batch.crop_window(30 * fs, aud, fs, 77, None, True, 2.5) == (40001, 1, 77)
    try:
        batch.crop_window(30 * fs, aud, fs, 77, None, True, None)
        assert False
    except IndexError:
        pass
