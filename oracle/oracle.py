"""ctypes/numpy front-end of the CPU oracle (oracle/xm_oracle.c).

TEST INFRASTRUCTURE ONLY: importable from tests/, bench.py's cpu_baseline leg and
__graft_entry__.smoke().  The product package never imports this module.

Arrays are numpy float32 in MATLAB layout: shape (H, W, C, N), Fortran order.
Function names and argument meaning follow the MATLAB operators they restate
(vl_nnconv, vl_nnpool, vl_nnbnorm, ... -- SURVEY.md section 8b).
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "libxm_oracle.so")
_lib = None

c_fp = C.POINTER(C.c_float)


def build(force=False):
    src = os.path.join(_HERE, "xm_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s", "clean", "all"])
    return _SO


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(_SO)
        _lib.orc_aud_samples.restype = C.c_double
        _lib.orc_aud_samples.argtypes = [C.c_int, C.c_double, C.c_double]
        _lib.orc_time2idx.argtypes = [C.c_double]
        # OpenMP's default team is one thread per LOGICAL cpu of the host (256 on the GPU boxes) whatever the container may
        # use: under the boxes' 16-core cgroup quota such a team is throttled for most of every period.  One thread per
        # physical core, no more than the quota pays for (see cpu_quota); set_num_threads() overrides.
        try:
            _lib.orc_set_num_threads(max(1, len(baseline_cpus())))
        except Exception:   # noqa: BLE001 -- an unreadable /proc or /sys must not break the checker
            pass
    return _lib


def F(a):
    """float32 Fortran-ordered copy/view (MATLAB `single`)."""
    return np.asfortranarray(a, dtype=np.float32)


def _p(a):
    return None if a is None else a.ctypes.data_as(c_fp)


def _shape4(a):
    s = tuple(a.shape) + (1,) * (4 - a.ndim)
    return [int(v) for v in s]


def _pad4(pad):
    if np.isscalar(pad):
        return [int(pad)] * 4
    pad = list(pad)
    if len(pad) == 1:
        return [int(pad[0])] * 4
    if len(pad) == 2:  # [py px] -> [t b l r]
        return [int(pad[0]), int(pad[0]), int(pad[1]), int(pad[1])]
    return [int(v) for v in pad]


def _pair(v):
    if np.isscalar(v):
        return [int(v), int(v)]
    v = list(v)
    return [int(v[0]), int(v[-1])]


def conv_out_size(n, pa, pb, f, d, s):
    return lib().orc_conv_out_size(int(n), int(pa), int(pb), int(f), int(d), int(s))


def vl_nnconv(x, f, b=None, dzdy=None, stride=1, pad=0, dilate=1, acc64=False,
              no_der_data=False, no_der_filters=False, no_der_biases=False):
    x, f = F(x), F(f)
    H, W, Cc, N = _shape4(x)
    FH, FW, FC, K = _shape4(f)
    sy, sx = _pair(stride)
    dy, dx = _pair(dilate)
    pt, pb, pl, pr = _pad4(pad)
    Ho = conv_out_size(H, pt, pb, FH, dy, sy)
    Wo = conv_out_size(W, pl, pr, FW, dx, sx)
    bb = None if b is None or np.size(b) == 0 else F(np.ravel(b))
    if dzdy is None:
        y = np.zeros((Ho, Wo, K, N), np.float32, order="F")
        rc = lib().orc_nnconv_forward(_p(x), H, W, Cc, N, _p(f), FH, FW, FC, K, _p(bb), _p(y),
                                      sy, sx, pt, pb, pl, pr, dy, dx, int(acc64))
        if rc:
            raise ValueError("vl_nnconv: bad shapes (rc=%d)" % rc)
        return y
    dzdy = F(dzdy)
    assert _shape4(dzdy) == [Ho, Wo, K, N], (dzdy.shape, (Ho, Wo, K, N))
    dxo = None if no_der_data else np.zeros(x.shape, np.float32, order="F")
    dfo = None if no_der_filters else np.zeros(f.shape, np.float32, order="F")
    dbo = None if (no_der_biases or bb is None) else np.zeros(K, np.float32)
    rc = lib().orc_nnconv_backward(_p(x), H, W, Cc, N, _p(f), FH, FW, FC, K, _p(dzdy), _p(dxo),
                                   _p(dfo), _p(dbo), sy, sx, pt, pb, pl, pr, dy, dx, int(acc64))
    if rc:
        raise ValueError("vl_nnconv: bad shapes (rc=%d)" % rc)
    return dxo, dfo, dbo


def vl_nnpool(x, pool, dzdy=None, stride=1, pad=0, method="max"):
    x = F(x)
    H, W, Cc, N = _shape4(x)
    ph, pw = _pair(pool)
    sy, sx = _pair(stride)
    pt, pb, pl, pr = _pad4(pad)
    m = {"max": 0, "avg": 1}[method]
    Ho = conv_out_size(H, pt, pb, ph, 1, sy)
    Wo = conv_out_size(W, pl, pr, pw, 1, sx)
    if dzdy is None:
        y = np.zeros((Ho, Wo, Cc, N), np.float32, order="F")
        rc = lib().orc_nnpool_forward(_p(x), H, W, Cc, N, ph, pw, sy, sx, pt, pb, pl, pr, m, _p(y))
        assert rc == 0
        return y
    dzdy = F(dzdy)
    dxo = np.zeros(x.shape, np.float32, order="F")
    rc = lib().orc_nnpool_backward(_p(x), H, W, Cc, N, ph, pw, sy, sx, pt, pb, pl, pr, m,
                                   _p(dzdy), _p(dxo))
    assert rc == 0
    return dxo


def vl_nnbnorm(x, g, b, dzdy=None, epsilon=1e-4, moments=None, acc64=False):
    """forward: returns (y, moments[C,2]); backward: (dx, dg, db, moments)."""
    x, g, b = F(x), F(np.ravel(g)), F(np.ravel(b))
    H, W, Cc, N = _shape4(x)
    mi = None if moments is None else F(np.reshape(moments, (Cc, 2), order="F"))
    mo = np.zeros((Cc, 2), np.float32, order="F")
    if dzdy is None:
        y = np.zeros(x.shape, np.float32, order="F")
        lib().orc_nnbnorm_forward(_p(x), H, W, Cc, N, _p(g), _p(b), C.c_float(epsilon), _p(mi),
                                  _p(y), _p(mo), int(acc64))
        return y, mo
    dzdy = F(dzdy)
    dxo = np.zeros(x.shape, np.float32, order="F")
    dg = np.zeros(Cc, np.float32)
    db = np.zeros(Cc, np.float32)
    lib().orc_nnbnorm_backward(_p(x), H, W, Cc, N, _p(g), _p(b), _p(dzdy), C.c_float(epsilon),
                               _p(mi), _p(dxo), _p(dg), _p(db), _p(mo), int(acc64))
    return dxo, dg, db, mo


def vl_nnrelu(x, dzdy=None, leak=0.0):
    x = F(x)
    y = np.zeros(x.shape, np.float32, order="F")
    d = None if dzdy is None else F(dzdy)
    lib().orc_nnrelu(_p(x), C.c_size_t(x.size), C.c_float(leak), _p(d), _p(y))
    return y


def vl_nnsigmoid(x, dzdy=None):
    x = F(x)
    y = np.zeros(x.shape, np.float32, order="F")
    d = None if dzdy is None else F(dzdy)
    lib().orc_nnsigmoid(_p(x), C.c_size_t(x.size), _p(d), _p(y))
    return y


def vl_nndropout(x, mask):
    """Y = vl_nndropout(X, 'mask', M) = M .* X (also the backward call: DZDX = M .* DZDY) [EXT MatConvNet]."""
    return np.asfortranarray((F(x) * F(mask)).astype(np.float32))


def dropout_mask(shape, rate, seed, offset=0):
    """The mask the PRODUCT draws (include/xmodal.h, xm_nndropout_forward): MATLAB's random stream cannot be
    reproduced, so there is no reference stream to follow; this restates the library's documented one -- Philox4x32-10,
    key = seed, counter = (column-major element index / 4 + offset), word e of the block for element 4 g + e,
    u = (word >> 8) / 2^24, mask = (u >= rate) / (1 - rate) -- so that tests can check it bit for bit."""
    n = int(np.prod(shape))
    g = np.arange((n + 3) // 4, dtype=np.uint64) + np.uint64(offset)
    c = [(g & np.uint64(0xFFFFFFFF)).astype(np.uint64), (g >> np.uint64(32)).astype(np.uint64),
         np.zeros_like(g), np.zeros_like(g)]
    k0, k1 = np.uint64(seed & 0xFFFFFFFF), np.uint64((seed >> 32) & 0xFFFFFFFF)
    M0, M1, m32 = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57), np.uint64(0xFFFFFFFF)
    for _ in range(10):
        p0, p1 = M0 * c[0], M1 * c[2]
        c = [((p1 >> np.uint64(32)) ^ c[1] ^ k0) & m32, p1 & m32, ((p0 >> np.uint64(32)) ^ c[3] ^ k1) & m32, p0 & m32]
        k0, k1 = (k0 + np.uint64(0x9E3779B9)) & m32, (k1 + np.uint64(0xBB67AE85)) & m32
    words = np.stack(c, 1).reshape(-1)[:n]
    u = (words >> np.uint64(8)).astype(np.float32) * np.float32(1.0 / 16777216.0)
    scale = np.float32(1.0) / (np.float32(1.0) - np.float32(rate))
    return np.asfortranarray(np.where(u >= np.float32(rate), scale, np.float32(0)).astype(np.float32).reshape(shape, order="F"))


def sum2(a, b, relu=False):
    a, b = F(a), F(b)
    y = np.zeros(a.shape, np.float32, order="F")
    lib().orc_sum2(_p(a), _p(b), C.c_size_t(a.size), int(relu), _p(y))
    return y


def scale_axpy(x, a, r=None, relu=False):
    x, a = F(x), F(np.ravel(a, order="F"))
    H, W, Cc, N = _shape4(x)
    rr = None if r is None else F(r)
    y = np.zeros(x.shape, np.float32, order="F")
    lib().orc_scale_axpy(_p(x), C.c_size_t(H * W), C.c_size_t(Cc * N), _p(a), _p(rr), int(relu),
                         _p(y))
    return y


def scale_backward(x, a, dzdy):
    x, a, dzdy = F(x), F(np.ravel(a, order="F")), F(dzdy)
    H, W, Cc, N = _shape4(x)
    dx = np.zeros(x.shape, np.float32, order="F")
    da = np.zeros((1, 1, Cc, N), np.float32, order="F")
    lib().orc_scale_backward(_p(x), C.c_size_t(H * W), C.c_size_t(Cc * N), _p(a), _p(dzdy),
                             _p(dx), _p(da))
    return dx, da


def vl_nnsoftmaxt(x, temperature=1.0):
    x = F(x)
    H, W, Cc, N = _shape4(x)
    y = np.zeros(x.shape, np.float32, order="F")
    lib().orc_nnsoftmaxt(_p(x), C.c_size_t(H * W), Cc, N, C.c_float(temperature), _p(y))
    return y


def vl_nnregloss(x, t, dzdy=None, kind="euclidean", sigma=1.0, instance_weights=None):
    """vl_nneuclideanloss / vl_nnhuberloss (mcnExtraLayers [EXT]); x, t: ... x N (last axis = sample)."""
    x, t = F(x), F(t)
    N = int(x.shape[-1]) if x.ndim == 4 else 1
    E = x.size // N
    w = None if instance_weights is None else F(np.ravel(instance_weights))
    k = {"euclidean": 0, "huber": 1}[kind]
    if dzdy is None:
        y = np.zeros(1, np.float32)
        lib().orc_nnregloss(_p(x), _p(t), C.c_size_t(E), N, k, C.c_float(sigma), _p(w), None, _p(y))
        return y[0]
    d = F(np.ravel(dzdy))
    y = np.zeros(x.shape, np.float32, order="F")
    lib().orc_nnregloss(_p(x), _p(t), C.c_size_t(E), N, k, C.c_float(sigma), _p(w), _p(d), _p(y))
    return y


def vl_nnsoftmaxt_backward(x, dzdy, temperature=1.0):
    x, dzdy = F(x), F(dzdy)
    H, W, Cc, N = _shape4(x)
    dx = np.zeros(x.shape, np.float32, order="F")
    lib().orc_nnsoftmaxt_backward(_p(x), _p(dzdy), C.c_size_t(H * W), Cc, N, C.c_float(temperature), _p(dx))
    return dx


def vl_nnsoftmaxceloss(x, p, dzdy=None, temperature=1.0, logit_targets=False,
                       instance_weights=None):
    x, p = F(x), F(p)
    H, W, Cc, N = _shape4(x)
    assert H == 1 and W == 1
    w = None if instance_weights is None else F(np.ravel(instance_weights))
    if dzdy is None:
        y = np.zeros(1, np.float32)
        lib().orc_nnsoftmaxceloss(_p(x), _p(p), Cc, N, C.c_float(temperature), int(logit_targets),
                                  _p(w), None, _p(y))
        return y[0]
    d = F(np.ravel(dzdy))
    y = np.zeros(x.shape, np.float32, order="F")
    lib().orc_nnsoftmaxceloss(_p(x), _p(p), Cc, N, C.c_float(temperature), int(logit_targets),
                              _p(w), _p(d), _p(y))
    return y


def vl_nnloss(x, c, dzdy=None, loss="softmaxlog"):
    x = F(x)
    H, W, Cc, N = _shape4(x)
    assert H == 1 and W == 1
    lab = F(np.ravel(c))
    lid = {"softmaxlog": 0, "classerror": 1}[loss]
    if dzdy is None:
        y = np.zeros(1, np.float32)
        lib().orc_nnloss(_p(x), _p(lab), Cc, N, lid, None, _p(y))
        return y[0]
    d = F(np.ravel(dzdy))
    y = np.zeros(x.shape, np.float32, order="F")
    lib().orc_nnloss(_p(x), _p(lab), Cc, N, lid, _p(d), _p(y))
    return y


def sgd_update(w, m, der, lr, momentum=0.9, wd=5e-4, batch=1.0):
    """in-place on copies; returns (w, m)."""
    w, m, der = F(w).copy(order="F"), F(m).copy(order="F"), F(der)
    lib().orc_sgd_update(_p(w), _p(m), _p(der), C.c_size_t(w.size), C.c_float(lr),
                         C.c_float(momentum), C.c_float(wd), C.c_float(batch))
    return w, m


def average_update(w, der, lr, nworkers=1.0):
    w, der = F(w).copy(order="F"), F(der)
    lib().orc_average_update(_p(w), _p(der), C.c_size_t(w.size), C.c_float(lr), C.c_float(nworkers))
    return w


def spec_rownorm(spec):
    s = F(spec)
    if s.ndim == 2:
        s = s.reshape(s.shape + (1,), order="F")
    H, W, N = s.shape[0], s.shape[1], int(np.prod(s.shape[2:]))
    out = np.zeros(s.shape, np.float32, order="F")
    lib().orc_spec_rownorm(_p(s), H, W, N, _p(out))
    return out.reshape(np.shape(spec), order="F")


def run_spec(z, fs=16000, Tw=25, Ts=10, alpha=0.97, nfft=1024):
    """runSpec [EXT, a-nagrani/VGGVox, un-vendored; restated from the VGGVox paper]: pre-emphasis
    y[t] = z[t] - alpha z[t-1], frames of Nw = Tw ms every Ns = Ts ms (no padding), symmetric Hamming
    window, |FFT_nfft|, bins 1..nfft/2 kept (512 x W).  z: L or L x N samples.  float64 throughout."""
    z = np.asarray(z, np.float64)
    if z.ndim == 1:
        z = z[:, None]
    Nw, Ns = int(round(1e-3 * Tw * fs)), int(round(1e-3 * Ts * fs))
    y = z.copy()
    y[1:] -= alpha * z[:-1]
    W = (z.shape[0] - Nw) // Ns + 1
    win = 0.54 - 0.46 * np.cos(2 * np.pi * np.arange(Nw) / (Nw - 1))
    out = np.zeros((nfft // 2, W, 1, z.shape[1]), np.float32, order="F")
    for n in range(z.shape[1]):
        fr = np.stack([y[j * Ns:j * Ns + Nw, n] * win for j in range(W)], 1)   # Nw x W
        out[:, :, 0, n] = np.abs(np.fft.fft(fr, nfft, axis=0))[:nfft // 2]
    return out


def resample(x, p, q, N=10, beta=5.0):
    """y = resample(x, p, q) of the speed-perturbation branch (getBatchEmoVoxCeleb.m:108).  [EXT] MATLAB Signal
    Processing Toolbox -- not in the reference tree, restated from its documented recipe (doubly unpinned, like
    run_spec): p, q reduced by gcd; h = firls(2 N max(p,q), [0 2fc 2fc 1], [1 1 0 0]) .* kaiser(L, 5), fc = 1/(2 max(p,q))
    -- a zero-width transition band makes the least-squares design the truncated ideal low-pass --, h <- p h / sum(h),
    zeros in front so that the delay is a whole number of output samples, y = upfirdn(x, h, p, q) with the delay removed
    and ceil(Lx p / q) samples kept.  float64 throughout; scipy.signal.resample_poly is the second opinion
    (tests/test_oracle.py)."""
    import math
    x = np.asarray(x, np.float64).ravel()
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
    Lx = x.size
    Ly = -(-Lx * p // q)
    y = np.zeros(Ly)
    t0 = (np.arange(Ly, dtype=np.int64) + delay) * q
    kmax = int(h.size // p) + 2
    khi = t0 // p
    for d in range(kmax):                      # y[j] = sum_k h[t0 - k p] x[k]
        k = khi - d
        t = t0 - k * p
        ok = (k >= 0) & (k < Lx) & (t >= 0) & (t < h.size)
        y[ok] += h[t[ok]] * x[k[ok]]
    return y


def time2idx(t):
    return lib().orc_time2idx(float(t))


def aud_samples(width, Tw_ms=25.0, fs=16000.0):
    return lib().orc_aud_samples(int(width), float(Tw_ms), float(fs))


def aggregate_logits(lg, first, last, agg="max"):
    lg = F(lg)
    Fr, E = lg.shape
    out = np.zeros(E, np.float32)
    lib().orc_aggregate_logits(_p(lg), Fr, E, int(first), int(last), {"max": 0, "mean": 1}[agg],
                               _p(out))
    return out


def crop_resize_face(src, avg3, image_size=(224, 224), crop=1 / 1.6):
    src = F(src)
    Hin, Win, c3, N = _shape4(src)
    a = F(np.ravel(avg3)[:3])
    out = np.zeros((image_size[0], image_size[1], 3, N), np.float32, order="F")
    lib().orc_crop_resize_face(_p(src), Hin, Win, N, C.c_double(crop), int(image_size[0]), int(image_size[1]),
                               _p(a), _p(out))
    return out


def normalize_face(rgb, avg3):
    rgb = F(rgb)
    H, W, c3, N = _shape4(rgb)
    assert c3 == 3
    out = np.zeros((H, W, 3, N), np.float32, order="F")
    a = F(np.ravel(avg3))
    lib().orc_normalize_face(_p(rgb), H, W, N, _p(a), _p(out))
    return out


def num_threads():
    return lib().orc_num_threads()


def physical_cores():
    """cores this process may run on, SMT siblings counted once (an OpenMP team on both hyper-threads of a core
    runs the SGEMM slower than one thread per core)"""
    try:
        allowed = os.sched_getaffinity(0)
    except AttributeError:
        allowed = set(range(os.cpu_count() or 1))
    seen, cur = set(), {}
    try:
        for line in open("/proc/cpuinfo"):
            if ":" not in line:
                if "processor" in cur and int(cur["processor"]) in allowed:
                    seen.add((cur.get("physical id", "0"), cur.get("core id", cur["processor"])))
                cur = {}
                continue
            k, v = line.split(":", 1)
            cur[k.strip()] = v.strip()
    except OSError:
        return len(allowed)
    return len(seen) or len(allowed)


def physical_core_cpus():
    """one logical CPU id per physical core this process may run on (the first SMT sibling)"""
    try:
        allowed = os.sched_getaffinity(0)
    except AttributeError:
        return list(range(os.cpu_count() or 1))
    first, cur = {}, {}
    try:
        for line in open("/proc/cpuinfo"):
            if ":" not in line:
                if "processor" in cur and int(cur["processor"]) in allowed:
                    first.setdefault((cur.get("physical id", "0"), cur.get("core id", cur["processor"])),
                                     int(cur["processor"]))
                cur = {}
                continue
            k, v = line.split(":", 1)
            cur[k.strip()] = v.strip()
    except OSError:
        pass
    return sorted(first.values()) or sorted(allowed)


def cpu_quota(root="/sys/fs/cgroup"):
    """CPU bandwidth limit of this process' cgroup in cores (cgroup v2 cpu.max, v1 cpu.cfs_quota_us / cpu.cfs_period_us),
    or None when there is none.  A team of more threads than the quota pays for is throttled by the kernel for part
    of every period -- pinned threads then stall in turn and a timed baseline swings by 2-3 x between boxes."""
    def read(path):
        try:
            return open(path).read().split()
        except OSError:
            return None
    v = read(root + "/cpu.max")
    if v and len(v) == 2 and v[0] != "max":
        try:
            return float(v[0]) / float(v[1])
        except ValueError:
            return None
    for d in (root + "/cpu", root + "/cpu,cpuacct"):
        q, per = read(d + "/cpu.cfs_quota_us"), read(d + "/cpu.cfs_period_us")
        if q and per:
            try:
                if float(q[0]) > 0:
                    return float(q[0]) / float(per[0])
            except ValueError:
                pass
    return None


def baseline_cpus():
    """the CPUs a timed baseline runs on: one per physical core of the affinity mask, no more of them than the
    cgroup's CPU quota pays for (min(affinity, quota))"""
    cpus = physical_core_cpus()
    q = cpu_quota()
    if q is not None:
        cpus = cpus[:max(1, min(len(cpus), int(q)))]
    return cpus


def set_num_threads(n=None):
    lib().orc_set_num_threads(int(n or len(baseline_cpus())))
    return num_threads()


def pin_threads():
    """one OpenMP thread per physical core (at most the cgroup's CPU quota of them), each pinned to its core (timed
    baselines); returns (threads, restore): call restore() afterwards -- the calling thread is team member 0 and was
    pinned as well."""
    cpus = baseline_cpus()
    try:
        before = os.sched_getaffinity(0)
    except AttributeError:
        before = None
    arr = (C.c_int * len(cpus))(*cpus)
    lib().orc_pin_threads(arr, len(cpus))

    def restore():
        if before is not None:
            os.sched_setaffinity(0, before)
    return num_threads(), restore
