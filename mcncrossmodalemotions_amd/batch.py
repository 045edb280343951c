"""Batch providers: getBatchEmoVoxCeleb (student) and getImageBatch (teacher) mirrors.

The reference's providers read wav / jpeg files (emoVoxCeleb/getBatchEmoVoxCeleb.m:76-194,
emoVoxCeleb/fetch_emovoxceleb_imdb.m:152-193).  File I/O and the FFT front-end (VGGVox runSpec)
are outside the built path; what is kept is every piece of arithmetic that
shapes the tensors the hot path consumes, fed from seeded synthetic sources:

    audSamp / crop window        getBatchEmoVoxCeleb.m:67-68,109-119
    time2idx + logit slicing     getBatchEmoVoxCeleb.m:145-158,210-214
    aggregation (max | mean)     getBatchEmoVoxCeleb.m:179-188      -> HIP xm_aggregate_logits
    per-row normalisation ('I')  getBatchEmoVoxCeleb.m:164-169      -> HIP xm_spec_rownorm
    speed perturbation ('S')     getBatchEmoVoxCeleb.m:102-108,217-245 -> HIP xm_resample (filter designed on the host)
    additive noise ('N')         getBatchEmoVoxCeleb.m:123-135      -> HIP xm_scale_axpy (z + Nratio * y)
    target selection + maxLabel  getBatchEmoVoxCeleb.m:30-32
    face normalisation           fetch_emovoxceleb_imdb.m:176-193   -> HIP xm_normalize_face
"""
import math

import numpy as np
import torch

from . import vl

FPS, LOGIT_STRIDE = 25, 6  # getBatchEmoVoxCeleb.m:212


def time2idx(t):
    """getBatchEmoVoxCeleb.m:210-214."""
    return int(math.floor(max(t * FPS - 1, 0) / LOGIT_STRIDE)) + 1


def aud_samples(width, Tw=25, fs=16000):
    """getBatchEmoVoxCeleb.m:67-68: audSamp = (0.01*W + 0.001*Tw - 0.001) * fs."""
    return (0.01 * width + 0.001 * Tw - 0.001) * fs


_SPEC_BANKS = {}


def _spec_filter_bank(fs, Tw, Ts, alpha, nfft, device):
    """1 x (Nw+1) x 1 x 2B filter bank = pre-emphasis * Hamming window * DFT rows (Re | Im), built in
    float64 on the host once per setting.  Tap k multiplies sample s[Ns*j - 1 + k]."""
    key = (fs, Tw, Ts, alpha, nfft, str(device))
    if key not in _SPEC_BANKS:
        Nw = int(round(1e-3 * Tw * fs))
        B = nfft // 2
        t = np.arange(Nw)
        win = 0.54 - 0.46 * np.cos(2 * np.pi * t / (Nw - 1))
        ang = 2 * np.pi * np.outer(t, np.arange(B)) / nfft            # Nw x B
        re, im = win[:, None] * np.cos(ang), -win[:, None] * np.sin(ang)
        bank = np.zeros((Nw + 1, 2 * B))
        bank[1:, :B] += re
        bank[1:, B:] += im
        bank[:-1, :B] -= alpha * re                                     # y[t] = s[t] - alpha s[t-1]
        bank[:-1, B:] -= alpha * im
        f = np.asfortranarray(bank.reshape(1, Nw + 1, 1, 2 * B).astype(np.float32))
        _SPEC_BANKS[key] = vl.from_numpy(f, device)
    return _SPEC_BANKS[key]


def runSpec(z, audio=None):
    """SPEC = runSpec(z, audio) on the device (getBatchEmoVoxCeleb.m:162, compute_audio_feats.m:176;
    [EXT] VGGVox, restated from the paper: 25 ms Hamming frames every 10 ms, pre-emphasis 0.97,
    1024-point FFT magnitude, 512 bins).  z: L x N device tensor of samples, one clip per column
    (column-major).  The whole STFT is ONE strided 1-D convolution on the MFMA path -- pre-emphasis,
    window and DFT folded into a 1 x 401 x 1 x 1024 filter bank, stride [1 160] -- followed by a
    magnitude/transposition kernel.  Returns 512 x W x 1 x N, W = floor((L - 400) / 160) + 1."""
    a = dict(fs=16000, Tw=25, Ts=10, alpha=0.97)
    a.update({k: v for k, v in (audio or {}).items() if k in a})
    nfft = 1024
    if z.dim() == 1:
        z = z[:, None]
    L, N = int(z.shape[0]), int(z.shape[1])
    Nw, Ns = int(round(1e-3 * a["Tw"] * a["fs"])), int(round(1e-3 * a["Ts"] * a["fs"]))
    if L < Nw:
        raise ValueError("runSpec: clip shorter than one analysis frame")
    bank = _spec_filter_bank(a["fs"], a["Tw"], a["Ts"], a["alpha"], nfft, z.device)
    # one leading zero per clip = the missing predecessor of the first sample (filter([1 -alpha], 1, z))
    buf = torch.zeros((N, L + 1), dtype=torch.float32, device=z.device)
    buf[:, 1:].copy_(z.t())
    x = buf.t()[None, :, None, :]                 # 1 x (L+1) x 1 x N view, column-major
    reim = vl.vl_nnconv(x, bank, None, stride=(1, Ns))
    return vl.spec_magnitude(reim)


def findSettings(transformation):
    """[chspeed, inputnorm, noisy] = findSettings(transformation, opts) -- getBatchEmoVoxCeleb.m:217-245: 'S' = speed
    perturbation, 'I' = per-row input normalisation, 'N' = additive noise; 'v' (validation) switches S and N off."""
    isVal = "v" in transformation
    return ("S" in transformation and not isVal), ("I" in transformation), ("N" in transformation and not isVal)


def resample_design(p, q, N=10, beta=5.0):
    """Filter of y = resample(x, p, q) [EXT: MATLAB Signal Processing Toolbox, called at getBatchEmoVoxCeleb.m:108;
    restated from its documented recipe]: p, q reduced by their gcd; h = firls(2 N max(p, q), [0 2fc 2fc 1], [1 1 0 0])
    .* kaiser(L, 5) with fc = 1 / (2 max(p, q)) -- with a zero-width transition band the least-squares design IS the
    truncated ideal low-pass 2 fc sinc(2 fc (n - (L-1)/2)) -- scaled to p * h / sum(h); zeros are put in front so that
    the delay is a whole number of OUTPUT samples, which is then dropped.  Returns (h float64, p, q, delay); the output
    has ceil(Lx p / q) samples."""
    g = math.gcd(int(p), int(q))
    p, q = int(p) // g, int(q) // g
    pqmax = max(p, q)
    fc = 0.5 / pqmax
    L = 2 * N * pqmax + 1
    n = np.arange(L, dtype=np.float64) - (L - 1) / 2
    h = 2 * fc * np.sinc(2 * fc * n) * np.kaiser(L, beta)
    h = p * h / h.sum()
    Lhalf = (L - 1) / 2
    nz = int(math.floor(q - (Lhalf % q)))
    h = np.concatenate([np.zeros(nz), h])
    delay = int(math.floor(math.ceil(Lhalf + nz) / q))
    return h, p, q, delay


def resample(x, p, q):
    """z = resample(zo, p, q) on the device (one clip, 1-D tensor)."""
    h, p, q, delay = resample_design(p, q)
    Ly = -(-int(x.numel()) * p // q)
    hd = torch.from_numpy(h.astype(np.float32)).to(x.device)
    return vl.resample(x.contiguous(), hd, p, q, delay, Ly)


class SyntheticEmoVoxImdb:
    """Stand-in for the imdb of fetch_emovoxceleb_imdb: per-track wav length (samples) and the
    cached teacher logits imdb.wavLogits{i} (F_i x 8 single, one row per sampled face frame)."""

    def __init__(self, num_tracks=64, seed=0, min_seconds=4.5, max_seconds=9.0, num_emotions=8, fs=16000,
                 val_fraction=0.0):
        rng = np.random.default_rng(seed)
        self.fs = fs
        self.num_samples = rng.integers(int(min_seconds * fs), int(max_seconds * fs), num_tracks)
        self.wavLogits = []
        for n in self.num_samples:
            frames = time2idx(n / fs)
            self.wavLogits.append(np.asfortranarray(rng.standard_normal((frames, num_emotions)).astype(np.float32) * 3))
        self.set = np.ones(num_tracks, int)       # imdb.images.set: 1 = train, 2 = val
        if val_fraction > 0:
            self.set[rng.permutation(num_tracks)[:int(round(num_tracks * val_fraction))]] = 2
        self.seed = seed
        self._dev = None

    def device_wav(self, ii, device):
        """synthetic waveform of track ii (seeded noise, num_samples[ii] samples) on the device."""
        cache = self.__dict__.setdefault("_wav", {})
        if ii not in cache:
            g = torch.Generator(device=device)
            g.manual_seed(self.seed * 100003 + int(ii))
            cache[ii] = torch.randn(int(self.num_samples[ii]), generator=g, device=device, dtype=torch.float32) * 0.1
        return cache[ii]

    # meta.noise of the reference (noisedir with `noisenum` wav files of `noiselen` samples, mixing volume `noisevol`,
    # getBatchEmoVoxCeleb.m:125-133): a seeded synthetic bank
    noisenum, noiselen, noisevol = 4, 20 * 16000, 0.3

    def device_noise(self, ir, device):
        cache = self.__dict__.setdefault("_noise", {})
        if ir not in cache:
            g = torch.Generator(device=device)
            g.manual_seed(self.seed * 7919 + 1000003 + int(ir))
            cache[ir] = torch.randn(self.noiselen, generator=g, device=device, dtype=torch.float32) * 0.05
        return cache[ir]

    def device_logits(self, device):
        """all tracks' logits concatenated (F_total x E) on the device + row offsets."""
        if self._dev is None:
            offs = np.cumsum([0] + [l.shape[0] for l in self.wavLogits])
            cat = np.asfortranarray(np.concatenate(self.wavLogits, 0))
            self._dev = (vl.from_numpy(cat, device), offs)
        return self._dev


def crop_window(total_samples, audSamp, fs, num_logit_rows, rng, fixedSegments=False, timeOffset=None):
    """(wr, startIdx, endIdx) of one clip, all 1-based as in cnn_get_batch_wav_emo (getBatchEmoVoxCeleb.m:81-152):
    wr is the first sample audioread takes, [startIdx, endIdx] the rows of the cached logits that are aggregated.
      random crop (:109-119):  total = min(19.9 fs, total) (:81-89);  wd = total - audSamp;
                               wd >= 1: wr = randi(wd) in [1, wd];  else wr = 1 (short clip, zero padded)
                               starttime = wr / fs, endtime = (wr + audSamp - 1) / fs (:141-142) -- wr enters 1-BASED --
                               rows time2idx(starttime) .. min(time2idx(endtime), #rows) (:145-152)
      fixedSegments (:91-101): wr = timeOffset * fs + 1, every row of the clip's logits (:136-137)."""
    if fixedSegments:
        if timeOffset is None:
            raise IndexError("fixedSegments: timeOffsets is empty (Index exceeds matrix dimensions upstream, :92)")
        return int(round(timeOffset * fs)) + 1, 1, int(num_logit_rows)
    total = min(int(total_samples), int(19.9 * fs))
    wd = total - int(round(audSamp))
    wr = int(rng.integers(1, wd + 1)) if wd >= 1 else 1
    s, e = time2idx(wr / fs), time2idx((wr + audSamp - 1) / fs)
    return wr, s, min(e, int(num_logit_rows))


def getBatchEmoVoxCeleb(imdb, batch, imageSize=(512, 300), numPredEmotions=8, logitAggregator="max",
                        lossType="hot-cross-ent", transformation="I", rng=None, spec_source=None,
                        device=None, use_wav=False, fixedSegments=False, timeOffsets=None):
    """inputs = getBatchEmoVoxCeleb(imdb, batch, ...) -> ['data', im, 'logitTarget', lgo,
    'maxLabel', maxLabel] (getBatchEmoVoxCeleb.m:31-43).  Spectrogram magnitudes come from
    `spec_source` (H x W x 1 x N device tensor), from the imdb's waveforms through the device
    front-end (`use_wav`: crop [wr, wr+audSamp) with zero padding of short clips :109-119, runSpec
    :162), or from a seeded half-normal generator.
    `fixedSegments` (:91-101, :136-137; off upstream, run_distillation.m:86): the crop of clip k starts at
    timeOffsets[k] seconds, clips are not thresholded to DATASET_LIMIT, and ALL cached logit rows of the clip are
    aggregated.  (Upstream always passes timeOffsets = [] (:15), so the branch cannot run there; an offset list is
    required here.)"""
    device = device or torch.device("cuda", torch.cuda.current_device())
    rng = rng or np.random.default_rng(0)
    batch = list(batch)
    N = len(batch)
    H, W = imageSize
    audSamp = aud_samples(W)
    chspeed, _, noisy = findSettings(transformation)
    if (chspeed or noisy) and not (use_wav and spec_source is None):
        raise ValueError("transformations 'S' / 'N' act on the waveform: they need use_wav=True")
    logits, offs = imdb.device_logits(device)
    first = np.zeros(N, np.int32)
    last = np.zeros(N, np.int32)
    crops = []
    for k, ii in enumerate(batch):
        total, rows = int(imdb.num_samples[ii]), imdb.wavLogits[ii].shape[0]
        speedR = audSampR = None
        if chspeed and not fixedSegments:
            # :102-108 -- draw order as upstream: speed first, then the crop offset; the window read is audSampR long
            # while the logit rows still follow [wr, wr + audSamp) (:141-142 use audSamp)
            total = min(total, int(19.9 * imdb.fs))
            speedR = 0.95 + float(rng.random()) * 0.1
            audSampR = int(round(audSamp * speedR))
            wd = total - audSampR
            if wd < 1:
                raise ValueError("clip %d is shorter than the speed-perturbed window (randi(wd) fails upstream, :106)" % ii)
            wr = int(rng.integers(1, wd + 1))
            s = time2idx(wr / imdb.fs)
            e = min(time2idx((wr + audSamp - 1) / imdb.fs), int(rows))
        else:
            # getBatchEmoVoxCeleb.m:81-89: no clip of the dataset is longer than DATASET_LIMIT = 19.9 s; the sample
            # count is thresholded accordingly (the cached teacher logits end there too)
            wr, s, e = crop_window(total, audSamp, imdb.fs, rows, rng, fixedSegments,
                                   None if timeOffsets is None else timeOffsets[k])
        mix = None
        if noisy:                                                       # :123-135, draw order Nir, Nwr, Nratio
            nz = (-(-audSampR * int(round(imdb.fs / speedR)) // imdb.fs)) if speedR else int(round(audSamp))
            mix = (int(rng.integers(1, imdb.noisenum + 1)), int(rng.integers(1, imdb.noiselen - nz + 1)),
                   float(rng.random()) * imdb.noisevol)
        crops.append((ii, wr - 1, speedR, audSampR, mix))   # 0-based slice start of audioread(audfile, [wr ...])
        first[k], last[k] = offs[ii] + s, offs[ii] + e
    if spec_source is None and use_wav:
        L = int(round(audSamp))
        z = torch.zeros((N, L), dtype=torch.float32, device=device)      # storage of the L x N mat
        for k, (ii, wr, speedR, audSampR, mix) in enumerate(crops):
            if speedR is not None:
                zo = imdb.device_wav(ii, device)[wr:wr + audSampR]
                w = resample(zo, int(round(imdb.fs / speedR)), imdb.fs)   # z = resample(zo, round(fs / speedR), fs)
                if abs(int(w.numel()) - L) > 160:
                    raise RuntimeError("resample produced %d samples for a window of %d" % (int(w.numel()), L))
                w = w[:L]        # (the spectrogram width only depends on floor((len - 400) / 160): +-1 sample is immaterial)
            else:
                w = imdb.device_wav(ii, device)[wr:wr + L]
            z[k, :w.numel()].copy_(w)                                    # zero padding when short (:117)
            if mix is not None:
                nir, nwr, ratio = mix
                # z + y .* Nratio runs over numel(z) (:128-134): a short clip was zero-padded to audSamp BEFORE the mix, so
                # its padded tail receives noise too; the resampled window of 'S' is not padded (its own length, cut to L)
                nzlen = int(w.numel()) if speedR is not None else L
                y = imdb.device_noise(nir, device)[nwr - 1:nwr - 1 + nzlen]
                a = vl.mat_empty(1, 1, 1, 1, device=device)
                a.fill_(ratio)
                zk = z[k, :nzlen]
                col = lambda t: t.reshape(1, 1, 1, -1).permute(3, 2, 1, 0)     # noqa: E731  (L x 1 x 1 x 1 mat view)
                zk.copy_(vl.scale_axpy(col(y), a, col(zk)).permute(3, 2, 1, 0).reshape(-1))   # z = z + y .* Nratio (:134)
        spec_source = runSpec(z.t(), {"fs": imdb.fs})
        if int(spec_source.shape[1]) != W:
            raise RuntimeError("runSpec produced %d frames, expected %d" % (int(spec_source.shape[1]), W))
    if spec_source is None:
        g = torch.Generator(device=device)
        g.manual_seed(int(rng.integers(0, 2 ** 31)))
        raw = torch.randn((N, 1, W, H), generator=g, device=device, dtype=torch.float32).abs_()
        spec_source = raw.permute(3, 2, 1, 0)
    im = vl.spec_rownorm(spec_source) if "I" in transformation else spec_source
    lgo, maxLabel = vl.aggregate_logits(logits, torch.from_numpy(first).to(device),
                                        torch.from_numpy(last).to(device), logitAggregator)
    if numPredEmotions > lgo.shape[2]:
        raise ValueError("numPredEmotions exceeds the number of cached logits")
    if numPredEmotions != lgo.shape[2]:
        # lgo = lgo(:,:,1:opts.numPredEmotions,:) ; [~, maxLabel] = max(lgo, [], 3)  (:30-32)
        lgo = lgo[:, :, :numPredEmotions, :].permute(3, 2, 1, 0).contiguous().permute(3, 2, 1, 0)
        maxLabel = vl.max_label(lgo)
    inputs = ["data", im]
    if lossType == "softmaxlog":
        inputs += ["maxLabel", maxLabel]
    elif lossType == "euclidean":
        weights = vl.mat_empty(1, 1, 1, N, device=device)   # "no re-weighting required" (:36)
        weights.fill_(1.0)
        inputs += ["logitTarget", lgo, "instanceWeights", weights, "maxLabel", maxLabel]
    elif lossType == "hot-cross-ent":
        inputs += ["logitTarget", lgo, "maxLabel", maxLabel]
    else:
        raise ValueError("unrecognised loss type: %s" % lossType)   # 'huber' included, as upstream (:41)
    return inputs


def getImageBatch(num, imageSize=(224, 224), averageImage=(131.0912, 103.8827, 91.4953), seed=1,
                  device=None, frameSize=None):
    """fetch_emovoxceleb_imdb.m:152-193 on synthetic frames: uint8-valued RGB -> rgb2gray ->
    replicate x3 -> minus the per-channel averageImage.  With `frameSize` = (Hin, Win) the frames are
    "decoded" at that size and go through the fused centre-crop(1/1.6) + bilinear-resize kernel first
    (the vl_imreadjpeg arguments of :161-167)."""
    device = device or torch.device("cuda", torch.cuda.current_device())
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    if frameSize is not None:
        raw = torch.randint(0, 256, (num, 3, frameSize[1], frameSize[0]), generator=g, device=device)
        return vl.crop_resize_face(raw.to(torch.float32).permute(3, 2, 1, 0), averageImage, imageSize)
    rgb = torch.randint(0, 256, (num, 3, imageSize[1], imageSize[0]), generator=g, device=device)
    rgb = rgb.to(torch.float32).permute(3, 2, 1, 0)
    return vl.normalize_face(rgb, averageImage)
