#!/usr/bin/env python
"""bench.py -- distillation-step throughput (face + audio pairs / s) on MI355X.

Metric (BASELINE.json): "distillation-step samples/sec (face+audio pair) at 1/2/4/8 MI355X".
One step = one pass of the hot path over one synthetic minibatch that is already resident in HBM:

  workload "distill" (default; BASELINE config 4 sharded as 32 pairs per GPU, weak scaling):
      frozen ResNet50 teacher forward (test-mode BN folded into the conv epilogues)
        -> logits -> logitTarget / maxLabel
      VGGVox student forward + soft-target CE (T = 2) + backward
      ParameterServer: one RCCL sum-all-reduce of the 66.6 MB gradient buffer (N > 1)
      SGD-momentum update (cnn_train_dag defaults), BN moments moving average
  workload "student"  : BASELINE config 2 (student fwd + bwd + update, 64 x 512x300, 1 GPU)
  workload "teacher"  : BASELINE config 3 (SE-ResNet50 teacher forward, batch 128, 1 GPU)
  workload "joint"    : BASELINE config 5 shard (SE-ResNet50 fwd+bwd + student fwd+bwd, 64 / GPU)

Prints ONE JSON line on rank 0 (contract in the task statement), with two extra objects:
  roofline     -- dominant convolution kernel: algorithmic FLOPs of its launches / their summed
                  durations, measured with HIP events on the launch stream in a second pass of
                  the same K steps (the first pass, without events, gives `value`);
  cpu_baseline -- the CPU oracle (oracle/, "port" of MatConvNet's CPU algorithm) timed on the
                  host cores over a bounded sample of the same workload (rank 0, N = 1 only).
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_FP32_MFMA_TFLOPS = 157.3  # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense, fp32 in/out

# algorithmic GFLOP per unit (conv + FC MACs x 2)
GFLOP = {"student_fwd_bwd_300": 16.633, "resnet50_fwd": 7.712, "senet50_fwd": 7.717,
         "senet50_fwd_bwd": 22.915}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1,
                    help="ranks = GPUs of this node.  Under torch.distributed.run (WORLD_SIZE set) it must equal the "
                         "world size; called directly with N > 1 the script starts the N ranks itself")
    ap.add_argument("--steps", type=int, default=0,
                    help="timed steps K (exactly K are timed).  0 = as many as fill --min-seconds, estimated from the "
                         "warm-up (the clocks of this part need ~0.3 s of sustained load to settle: a 0.2 s timed "
                         "region under-reports by 5-10 %)")
    ap.add_argument("--warmup", type=int, default=0,
                    help="untimed steps W; 0 = 10.  Untimed 'settle' steps follow until 1.5 s of load have passed")
    ap.add_argument("--min-seconds", type=float, default=2.0)
    ap.add_argument("--workload", default="distill",
                    choices=["distill", "student", "teacher", "joint", "cpu-teacher"])
    ap.add_argument("--per-gpu-batch", type=int, default=0)
    ap.add_argument("--width", type=int, default=300, help="spectrogram width (3 s clips)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--cpu-pairs", type=int, default=0, help="CPU baseline sample size (0 = scaled to the host: ~cores/16, "
                    "at least 8)")
    ap.add_argument("--parserv", default="auto", choices=["auto", "torch", "rccl-capi"],
                    help="gradient exchange: the library's own communicator behind the C ABI (xm_parserv_push / sync -- "
                         "what a MATLAB host binds; the default whenever an exchange happens) or torch.distributed")
    ap.add_argument("--teacher", default="resnet50", choices=["resnet50", "senet50"],
                    help="frozen teacher of the distill workload (BASELINE config 4 names resnet50)")
    ap.add_argument("--teacher-lanes", type=int, default=2,
                    help="teacher workload: sample slices evaluated concurrently on that many HIP streams")
    ap.add_argument("--wgrad-stream", type=int, default=1,
                    help="1: filter/bias derivatives of the student on a side HIP stream")
    ap.add_argument("--frames", type=int, default=1,
                    help="distill: face frames per pair; F > 1 runs the teacher on nb*F faces and max-aggregates "
                         "their logits per pair (getBatchEmoVoxCeleb.m:145-158,179-185; SURVEY 8f row 1)")
    ap.add_argument("--imdb-windows", type=int, default=0,
                    help="distill: 1 = RAGGED frame windows as getBatch draws them from an imdb (getBatchEmoVoxCeleb.m:137-158): "
                         "per pair a synthetic track of 4-20 s, logit rows every 6th frame at 25 fps (time2idx, :210-214), a "
                         "random 512 x W spectrogram window -> rows time2idx(start) .. min(time2idx(end), rows the track has "
                         "-- ffmpeg drops up to 3 of them on every 10th track); the teacher runs on exactly those frames.  "
                         "Replaces --frames")
    ap.add_argument("--overlap-allreduce", type=int, default=1,
                    help="1: fc6-8 gradient bucket all-reduced while the rest of the backward pass runs (N > 1)")
    ap.add_argument("--exec-hint", default="auto", choices=["auto", "0", "1"],
                    help="xm_set_exec_hint: 1 = XM_EXEC_SINGLE_STREAM (the library may pick kernels that assume nothing "
                         "else is resident), 0 = none; auto = 1 with --serial, 0 otherwise.  `--serial --exec-hint 0` "
                         "runs the kernels of the overlapped timed region one after the other (profiles/)")
    ap.add_argument("--serial", action="store_true",
                    help="one HIP stream, no overlap anywhere (what the roofline leg and the rocprof profile use: "
                         "kernel durations are then those of isolated kernels)")
    ap.add_argument("--teacher-batch", type=int, default=0,
                    help="distill: faces per frozen-teacher pass (a multiple of the per-GPU batch; 0 = the per-GPU "
                         "batch).  The reference decouples the two as well: buildImdb runs the teacher at batch 128 "
                         "(fetch_emovoxceleb_imdb.m:63), the student trains at 64 (run_distillation.m:75).  One pass "
                         "feeds teacher-batch / per-gpu-batch consecutive steps; every pair still gets its own face")
    ap.add_argument("--teacher-chunk", type=int, default=0,
                    help="distill: evaluate the frozen teacher in sample slices of this many faces, one after the other on "
                         "its stream (samples are independent in test mode; a slice's activations stay in the 256 MB "
                         "Infinity Cache between layers).  0 = the whole batch in one pass")
    ap.add_argument("--north-star", type=int, default=-1,
                    help="1: after the main measurement, time north_star's own configuration (SE-ResNet50 teacher + "
                         "VGGVox student, 256 pairs on this GPU) for a bounded number of steps in a child process and "
                         "attach it as `north_star_b256`.  -1 = on for the default single-GPU distill line only")
    ap.add_argument("--teacher-prefetch", type=int, default=1,
                    help="1: the teacher stream works one batch ahead of the student (needs --overlap-teacher 1)")
    ap.add_argument("--teacher-gate", default="",
                    help="distill + --teacher-prefetch: where in the student step the teacher pass of the NEXT step may "
                         "start on its stream: '' = at the start of the step; a student layer name (bn1, conv2, ...) = when "
                         "the backward pass reaches that layer -- the MFMA-bound teacher then runs next to the HBM-bound "
                         "tail of this step (first layer's bnorm / pooling derivative, update) and head of the next")
    ap.add_argument("--overlap-teacher", type=int, default=1,
                    help="1: the frozen teacher runs on a second HIP stream next to the student forward (+2 %%)")
    return ap.parse_args()


def pmc_traffic(kernel):
    """HBM bytes per launch of `kernel` from the committed rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes
    (profiles/rNN/pmc_traffic.json, written by tools/collect_profiles.sh; PMC counters cannot be
    collected from inside this process).  FETCH_SIZE is doubled per MI355X_MICROARCH.md (gfx950).
    Returns (bytes, file, commit the profile was taken at) -- the figure is a property of THAT build."""
    import glob
    root = os.path.dirname(os.path.abspath(__file__))
    files = sorted(glob.glob(os.path.join(root, "profiles", "r*", "pmc_traffic.json")))
    if not files:
        return None, None, None
    try:
        tab = json.load(open(files[-1]))
    except (OSError, ValueError):
        return None, None, None
    commit = tab.get("_meta", {}).get("commit") if isinstance(tab.get("_meta"), dict) else None
    for k, v in tab.items():
        if k.endswith(kernel) and isinstance(v, dict):
            return int(v["fetch_bytes_x2"] + v["write_bytes"]), os.path.relpath(files[-1], root), commit
    return None, os.path.relpath(files[-1], root), commit


def host_info():
    model = "unknown"
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                model = line.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    return model


def launch_ranks(args):
    """`python bench.py --gpus N` with N > 1 and no launcher around it: become `python -m torch.distributed.run
    --nnodes=1 --nproc-per-node N ... bench.py <same arguments>` (one process per GPU, cnn_train_dag's
    numel(opts.gpus) workers, run_distillation.m:71,77,179-181).  Refuses -- never runs fewer ranks than asked for --
    when the node shows fewer than N devices, unless XM_DEBUG_DIST=gloo0 (functional run of the N-rank path on one GPU)."""
    import socket
    import torch
    have = torch.cuda.device_count() if torch.cuda.is_available() else 0
    if os.environ.get("XM_DEBUG_DIST") != "gloo0" and have < args.gpus:
        raise SystemExit("bench.py --gpus %d: only %d GPU(s) visible on this node; refusing to run fewer ranks than "
                         "requested (XM_DEBUG_DIST=gloo0 runs all ranks on cuda:0 over gloo for a functional check)"
                         % (args.gpus, have))
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    sys.stdout.flush()
    sys.stderr.flush()
    os.execve(sys.executable, cmd, dict(os.environ))


def main():
    args = parse()
    if args.workload == "cpu-teacher":
        return cpu_teacher_line(args)
    if "WORLD_SIZE" not in os.environ:
        if args.gpus > 1:
            return launch_ranks(args)
    elif int(os.environ["WORLD_SIZE"]) != args.gpus:
        raise SystemExit("bench.py --gpus %d under a launcher with WORLD_SIZE=%s: the two must agree"
                         % (args.gpus, os.environ["WORLD_SIZE"]))
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC (RCCL across processes)
    # main + wgrad + teacher streams + RCCL's: more than the default 4 hardware queues, see the package
    # __init__ (without this the stream overlap is serialised as soon as a process group exists)
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
    import torch
    import torch.distributed as dist
    from mcncrossmodalemotions_amd import _lib, vl, zoo, train, batch as xbatch, dagnn

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X; there is no CPU fallback for the product path")
    # XM_DEBUG_DIST=gloo0: every rank on cuda:0, exchange over gloo -- a functional run of the N > 1 code path on a
    # one-GPU box (RCCL refuses two ranks on one device); the throughput it prints means nothing
    shared_gpu = os.environ.get("XM_DEBUG_DIST") == "gloo0"
    if shared_gpu:
        local = 0
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    force_dist = os.environ.get("XM_DEBUG_DIST") in ("1", "2", "3")   # exercise the RCCL path with a single rank
    # "3": the library's own communicator only, no torch process group next to it
    if args.parserv == "auto":
        args.parserv = "torch" if shared_gpu else "rccl-capi"      # gloo debug runs have no RCCL communicator
    # ONE RCCL communicator per rank: with the library's own communicator doing the exchange, the torch
    # process group is only the control plane -- rendezvous store, barrier, agreeing on K -- all host-side, so it
    # runs on gloo and never creates a second RCCL communicator next to the library's.  --parserv torch is the
    # other arrangement: torch's nccl group IS the exchange path and the library creates none.
    ctl_backend = "gloo" if (shared_gpu or args.parserv == "rccl-capi") else "nccl"
    if world > 1 or (force_dist and os.environ.get("XM_DEBUG_DIST") != "3"):
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        if ctl_backend == "gloo":
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    L = _lib.load()
    # HOW this host calls the library, stated once and before the first operator call (include/xmodal.h): the kernel a
    # shape gets depends on (shape, table, this hint) only -- the roofline leg below runs on one stream but keeps the
    # hint of the timed region, so both run the same kernels
    one_stream = args.serial or (not args.wgrad_stream and (args.workload in ("student", "joint") or not args.overlap_teacher))
    exec_hint = int(args.exec_hint) if args.exec_hint != "auto" else (1 if one_stream else 0)
    vl.set_exec_hint(vl.EXEC_SINGLE_STREAM if exec_hint else 0)

    def ctl_max(x, dtype):
        """MAX over the workers of one host number, through the control group (host tensors on gloo)"""
        t = torch.tensor([x], dtype=dtype, device=dev if ctl_backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return t.item()

    wl = args.workload
    nb = args.per_gpu_batch or {"distill": 32, "student": 64, "teacher": 128, "joint": 64}[wl]
    W = args.width
    seed = 4 + rank

    # ParameterServer FIRST -- before any network, buffer or side stream exists.  A communicator created after the
    # operator streams were in use cost 12 % of the step on this stack (3365 vs 3842 pairs/s at one rank, the whole
    # backward phase 0.8 ms longer even when no collective was ever issued; moving xm_comm_init here removed it).
    force_ps = bool(force_dist and os.environ.get("XM_DEBUG_DIST") in ("1", "3"))
    if os.environ.get("XM_PS_LATE"):         # experiment: communicator created AFTER the networks (the call-order trap)
        parserv = train.ParameterServer(args.parserv)
        parserv.force = force_ps
    else:
        # every worker ends up on the same backend: the library's communicator, or torch.distributed if any failed
        parserv = train.ParameterServer.start_agreed(args.parserv, force=force_ps)
        if parserv.backend == "torch" and args.parserv != "torch" and ctl_backend == "gloo" and not shared_gpu \
                and dist.is_initialized():
            # agreed fallback: the library's communicator is gone (destroyed on every worker), so torch's nccl group
            # is again the only RCCL communicator of the rank; the gloo group stays the control plane
            parserv.group = dist.new_group(backend="nccl", device_id=dev)
        args.parserv = parserv.backend
    # ---- networks ------------------------------------------------------------------------
    teacher = student = None
    if wl in ("distill", "teacher", "joint"):
        tname = ("%s-ferplus" % args.teacher) if wl == "distill" else "senet50-ferplus"
        teacher = zoo.ferPlusZoo(tname, seed=100 if wl == "distill" else 300)
        if wl == "joint":
            teacher.removeLayer("top1error")
            teacher.pack_params()
        else:
            zoo.strip_losses(teacher)  # fetch_emovoxceleb_imdb.m:101-106
            teacher.move("gpu")
            teacher.vars["prediction"].precious = True
    if wl in ("distill", "student", "joint"):
        student = zoo.emoVoxZoo("emovoxceleb-student", scratch=1, lossType="hot-cross-ent",
                                numSeconds=W / 100.0, numOutputs=8, seed=200)
        student.pack_params()
    parserv.start()
    parserv.overlap = bool(args.overlap_allreduce)
    opts = train.TrainOpts(batchSize=nb * world)

    # ---- synthetic inputs, resident in HBM before the timed region ------------------------
    faces = spec = lgo = lab = flab = None
    tmult = 1
    win = None
    if wl == "distill" and args.imdb_windows:
        win = imdb_windows(nb, W, seed)
    if teacher is not None:
        F = args.frames if wl == "distill" else 1
        if win is not None:
            F = 2                                         # (any value > 1: the multi-frame branch below; counts come from `win`)
        if wl == "distill" and F == 1 and args.teacher_batch:
            if args.teacher_batch % nb:
                raise SystemExit("--teacher-batch must be a multiple of the per-GPU batch")
            tmult = args.teacher_batch // nb
        faces = xbatch.getImageBatch(win["frames"] if win is not None else nb * F * tmult, seed=seed, device=dev)
        calib = xbatch.getImageBatch(min(nb, 16), seed=999, device=dev)
        zoo.calibrate_moments(teacher, ["data", calib])  # realistic stored moments
        teacher.mode = "test" if wl != "joint" else "normal"
        if wl == "joint":
            rng = np.random.default_rng(seed)
            flab = vl.from_numpy(rng.integers(1, 9, (1, 1, 1, nb)).astype(np.float32), dev)
    if student is not None:
        g = torch.Generator(device=dev)
        g.manual_seed(seed)
        raw = torch.randn((nb, 1, W, 512), generator=g, device=dev, dtype=torch.float32).abs_()
        spec = vl.spec_rownorm(raw.permute(3, 2, 1, 0))  # getBatchEmoVoxCeleb.m:164-169
        if wl == "student":
            rng = np.random.default_rng(seed)
            lgo = vl.from_numpy((rng.standard_normal((1, 1, 8, nb)) * 3).astype(np.float32), dev)
            lab = vl.max_label(lgo)

    F = args.frames if wl == "distill" else 1
    if win is not None:
        F = 2
    _tp = os.environ.get("XM_TEACHER_PRIO")
    tstream = (torch.cuda.Stream(device=dev, priority=int(_tp)) if _tp is not None else torch.cuda.Stream(device=dev)) \
        if (wl == "distill" and args.overlap_teacher and F == 1) else None
    frozen = zoo.FrozenTeacher(teacher, lanes=args.teacher_lanes) if (wl == "teacher" or F > 1) else None
    if win is not None:
        first = torch.tensor(win["first"], device=dev, dtype=torch.int32)      # 1-based, inclusive, ragged
        last = torch.tensor(win["last"], device=dev, dtype=torch.int32)
    elif F > 1:
        first = torch.arange(0, nb, device=dev, dtype=torch.int32) * F + 1   # 1-based, inclusive
        last = first + (F - 1)
    if args.wgrad_stream:
        _sp = os.environ.get("XM_SIDE_PRIO")
        side = torch.cuda.Stream(device=dev, priority=int(_sp)) if _sp is not None else torch.cuda.Stream(device=dev)
        if student is not None:
            student.wgradStream = side
        if wl == "joint":
            teacher.wgradStream = side

    prefetched = {}
    mode = {"serial": bool(args.serial)}
    side_streams = {id(n): n.wgradStream for n in (student, teacher) if n is not None}

    def set_serial(on):
        mode["serial"] = on
        prefetched.clear()
        for n in (student, teacher):
            if n is not None:
                n.wgradStream = None if on else side_streams[id(n)]

    set_serial(mode["serial"])

    def teacher_logits(x):
        """frozen teacher forward -> 1 x 1 x 8 x n logits (fetch_emovoxceleb_imdb.m:129-130), optionally in sample slices"""
        n, ch = int(x.shape[3]), args.teacher_chunk
        if not ch or ch >= n:
            teacher.eval(["data", x])
            return teacher.vars["prediction"].value
        out = vl.mat_empty(1, 1, 8, n, device=x.device)
        for a in range(0, n, ch):
            teacher.eval(["data", x[..., a:a + ch]])
            out[..., a:a + ch].copy_(teacher.vars["prediction"].value)
        return out

    def step(it):
        if wl == "teacher":
            if mode["serial"]:
                teacher.eval(["data", faces])
            else:
                frozen.logits(faces)   # fetch_emovoxceleb_imdb.m:129-130
            return
        if wl == "student":
            train.train_step(student, ["data", spec, "logitTarget", lgo, "maxLabel", lab], opts, it,
                             parserv, nb * world)
            return
        if wl == "distill" and F > 1:
            # multi-frame pair: teacher over all nb*F frames (two stream lanes), per-pair max over the
            # window's frames (getBatchEmoVoxCeleb.m:179-185), then the student step
            if mode["serial"]:
                teacher.eval(["data", faces])
                lg = teacher.vars["prediction"].value
            else:
                lg = frozen.logits(faces)                              # 1 x 1 x 8 x (nb*F)
            fl = lg.permute(3, 2, 1, 0).reshape(int(lg.shape[3]), 8).t().contiguous().t()   # frames x 8 mat
            tl, ml = vl.aggregate_logits(fl, first, last, "max")
            train.train_step(student, ["data", spec, "logitTarget", tl, "maxLabel", ml], opts, it,
                             parserv, nb * world)
            return
        if wl == "distill" and (tstream is None or mode["serial"]):
            tl = teacher_logits(faces if tmult == 1 else faces[..., :nb])      # 1 x 1 x 8 x nb teacher logits
            ml = vl.max_label(tl)                       # getBatchEmoVoxCeleb.m:32
            train.train_step(student, ["data", spec, "logitTarget", tl, "maxLabel", ml], opts, it,
                             parserv, nb * world)
            return
        if wl == "distill":
            # the frozen teacher runs on its own HIP stream; only the loss / metric layers of the
            # student wait for its logits (input_events).  With --teacher-prefetch the teacher works
            # one batch ahead, like getBatch's prefetch call (getBatchEmoVoxCeleb.m:20-24): every step
            # still launches exactly one teacher forward and one student step.
            main = torch.cuda.current_stream()

            def launch_teacher(gate=None):
                # prefetch: pipeline depth 1 -- the teacher of step i+1 starts no earlier than the
                # student of step i (event recorded on the main stream at the start of this step, or -- `gate` --
                # where its backward pass reached the --teacher-gate layer)
                if gate is not None:
                    tstream.wait_event(gate)
                elif args.teacher_prefetch:
                    e0 = torch.cuda.Event()
                    e0.record(main)
                    tstream.wait_event(e0)
                else:
                    tstream.wait_stream(main)
                with torch.cuda.stream(tstream):
                    tl_ = teacher_logits(faces)
                    ml_ = vl.max_label(tl_)
                    ev_ = torch.cuda.Event()
                    ev_.record(tstream)
                tl_.record_stream(main)
                ml_.record_stream(main)
                return tl_, ml_, ev_

            if args.teacher_prefetch:
                # queue of per-step logit slices; one teacher pass (tmult * nb faces) refills it when a single
                # step of slack is left, i.e. the teacher never runs more than one pass ahead
                q = prefetched.setdefault("q", [])

                def refill(gate=None):
                    tl_, ml_, ev_ = launch_teacher(gate)
                    for k in range(tmult):
                        q.append((tl_[..., k * nb:(k + 1) * nb], ml_[..., k * nb:(k + 1) * nb], ev_))
                if not q:
                    refill()
                tl, ml, ev = q.pop(0)
                if args.teacher_gate and tmult == 1:
                    # this step first (its backward records the gate event), then the next step's teacher pass behind it
                    gate = {}

                    def hook():
                        gate["ev"] = torch.cuda.Event()
                        gate["ev"].record(main)
                    student.bwdHooks = {args.teacher_gate: hook}
                    train.train_step(student, ["data", spec, "logitTarget", tl, "maxLabel", ml], opts, it,
                                     parserv, nb * world, input_events={"logitTarget": ev, "maxLabel": ev})
                    student.bwdHooks = {}
                    if "ev" not in gate:
                        raise SystemExit("--teacher-gate %r: no such layer in the student's backward pass" % args.teacher_gate)
                    refill(gate["ev"])
                    return
                if len(q) == 0:
                    refill()
            else:
                tl, ml, ev = launch_teacher()
                if tmult > 1:
                    tl, ml = tl[..., :nb], ml[..., :nb]
            train.train_step(student, ["data", spec, "logitTarget", tl, "maxLabel", ml], opts, it,
                             parserv, nb * world, input_events={"logitTarget": ev, "maxLabel": ev})
            return
        # joint: teacher fwd+bwd (hard-label CE head, ferPlusZoo.m:240-249) + student distillation
        teacher.vars["prediction"].precious = True
        train.train_step(teacher, ["data", faces, "label", flab], opts, it, parserv, nb * world)
        tl = teacher.vars["prediction"].value
        ml = vl.max_label(tl)
        train.train_step(student, ["data", spec, "logitTarget", tl, "maxLabel", ml], opts, it, parserv,
                         nb * world)

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    # Bounded run-ahead: the host enqueues a step in ~1.5-3 ms, the GPU needs 9-13 ms.  Left alone the
    # host fills the HIP queues; the runtime's queue-full path then stalls launches for milliseconds
    # while the GPU drains and idles (measured: student step 13.1 -> 15.2 ms in most processes).  Keeping
    # at most two steps in flight costs nothing and makes the step time reproducible.
    inflight = []

    host_t = [0.0, 0]

    def throttled_step(it):
        th = time.perf_counter()
        step(it)
        host_t[0] += time.perf_counter() - th
        host_t[1] += 1
        ev = torch.cuda.Event()
        ev.record()
        inflight.append(ev)
        if len(inflight) > 2:
            inflight.pop(0).synchronize()

    # diagnostics (XM_BENCH_MARKS=1): timing events on the main stream at the phase boundaries of the student step;
    # printed to stderr after the timed region (where the main stream spends its time, incl. the wait for the side
    # stream) -- a rocprofv3 trace cannot show this, its launch overhead makes the run host-bound
    marks_log = []
    if os.environ.get("XM_BENCH_MARKS") and student is not None:
        def _mark(label):
            e = torch.cuda.Event(enable_timing=True)
            e.record()
            marks_log.append((label, e))
        student.markHook = _mark
    _mp = os.environ.get("XM_MAIN_PRIO")
    if _mp is not None:
        torch.cuda.set_stream(torch.cuda.Stream(device=dev, priority=int(_mp)))
    # warm-up: Wm steps (default 10) and at least 1.5 s -- tile tuning happens here, and the shader clock of this
    # part takes ~0.3 s of sustained load to reach its steady state (tools/power_probe.py)
    Wm = args.warmup or 10
    tw = time.perf_counter()
    for it in range(Wm):
        throttled_step(it)
    # settle: the contract's W warm-up steps can be as short as 50 ms; keep stepping (untimed, reported as
    # "settle_steps") until the device has seen 1.5 s of sustained load (0.3 s for the clock; the first process on a
    # cold box also shows a one-off ~50 ms stall about 1.2 s after its first kernel), so that the K timed steps run at the
    # steady-state clock whatever W was
    barrier()
    t1 = time.perf_counter()
    for j in range(3):
        throttled_step(Wm + j)
    barrier()
    est = (time.perf_counter() - t1) / 3
    settle = 3
    more = max(0, int(np.ceil((1.5 - (time.perf_counter() - tw)) / max(est, 1e-6))))
    if world > 1:   # same count on every rank: each step contains collectives
        more = int(ctl_max(more, torch.int64))
    for j in range(more):
        throttled_step(Wm + settle + j)
    settle += more
    barrier()
    K = args.steps
    if K <= 0:
        # time EXACTLY K steps with K >= min_seconds / (step time estimated above)
        K = max(20, int(np.ceil(args.min_seconds / max(est, 1e-6))))
        if world > 1:   # every rank must time the same K
            K = int(ctl_max(K, torch.int64))
    args.steps, args.warmup = K, Wm
    Wm = Wm + settle   # step counter offset only
    # the timed region: exactly K steps between barrier + synchronize on both sides.  Window marks are HIP events
    # recorded on the main stream every K/8 steps (no host synchronisation): min / median / max window rate show
    # the spread inside the region.
    nwin = 8 if K >= 40 else (4 if K >= 16 else 1)   # the driver's --steps 20: four windows of five steps
    marks = []
    barrier()
    t0 = time.perf_counter()
    for it in range(K):
        if nwin > 1 and it % (K // nwin) == 0 and len(marks) < nwin:
            ev = torch.cuda.Event(enable_timing=True)
            ev.record()
            marks.append((it, ev))
        throttled_step(Wm + it)
    evl = torch.cuda.Event(enable_timing=True)
    evl.record()
    marks.append((K, evl))
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        dt = float(ctl_max(dt, torch.float64))
    ms_per_step = dt / args.steps * 1e3
    if marks_log:
        student.markHook = None
        seq = marks_log[-6 * min(args.steps, 40):]
        while seq and seq[0][0] != "fwd0":
            seq.pop(0)
        acc = {}
        for (la, ea), (lb, eb) in zip(seq, seq[1:]):
            k = "%s->%s" % (la, lb)
            acc.setdefault(k, []).append(ea.elapsed_time(eb))
        print("[marks] " + "  ".join("%s %.3f ms" % (k, float(np.mean(v))) for k, v in acc.items()) +
              "  host enqueue %.3f ms/step" % (host_t[0] / max(1, host_t[1]) * 1e3), file=sys.stderr)
    units = nb * world
    value = units * args.steps / dt
    windows = None
    if len(marks) > 2:
        rates = [nb * world * (marks[i + 1][0] - marks[i][0]) / (marks[i][1].elapsed_time(marks[i + 1][1]) * 1e-3)
                 for i in range(len(marks) - 1)]
        windows = {"n": len(rates), "steps_each": marks[1][0] - marks[0][0], "min": round(min(rates), 1),
                   "median": round(float(np.median(rates)), 1), "max": round(max(rates), 1),
                   "rates": [round(r, 1) for r in rates],
                   "note": "rank-0 stream time between event marks inside the timed region (value uses the wall clock)"}

    # ---- roofline leg: same K steps again with HIP events around every conv launch ---------
    roofline = None
    if not args.no_roofline:
        # serial: with the overlap streams on, kernels share the chip and a launch's duration is
        # not the kernel's own speed
        was_serial = mode["serial"]
        set_serial(True)
        # (the execution hint is unchanged: the kernels are those of the timed region)
        step(args.warmup + args.steps)   # shapes of the serial path (full-batch teacher) get tuned
        torch.cuda.synchronize()
        rsteps = min(args.steps, 60)
        for it in range(10):                       # clocks back to steady state in serial mode
            throttled_step(args.warmup + args.steps + 1 + it)
        torch.cuda.synchronize()
        L.xm_prof_enable(1)
        for it in range(rsteps):
            throttled_step(args.warmup + args.steps + 11 + it)
        torch.cuda.synchronize()
        L.xm_prof_enable(0)
        set_serial(was_serial)
        cap = 64
        keys = (C.c_int * cap)()
        ms = (C.c_double * cap)()
        fl = (C.c_double * cap)()
        cnt = (C.c_longlong * cap)()
        n = L.xm_prof_collect(cap, keys, ms, fl, cnt)
        bkeys = (C.c_int * cap)()
        by = (C.c_double * cap)()
        nb_ = L.xm_prof_collect_bytes(cap, bkeys, by)
        bytes_of = {bkeys[i]: by[i] for i in range(min(nb_, cap))}
        rows = []
        for i in range(min(n, cap)):
            buf = C.create_string_buffer(128)
            L.xm_prof_kernel_name(keys[i], buf, 128)
            rows.append({"kernel": buf.value.decode(), "ms": ms[i], "flops": fl[i], "launches": int(cnt[i]),
                         "bytes": bytes_of.get(keys[i], 0.0)})
        rows.sort(key=lambda r: -r["ms"])
        if rows:
            d = rows[0]
            ach = d["flops"] / (d["ms"] * 1e-3) / 1e12
            conv_ms = sum(r["ms"] for r in rows)
            conv_fl = sum(r["flops"] for r in rows)
            traffic, tsrc, tcommit = pmc_traffic(d["kernel"])
            roofline = {"bound": "mfma", "kernel": d["kernel"], "achieved": round(ach, 2),
                        "peak": PEAK_FP32_MFMA_TFLOPS, "unit": "TFLOP/s",
                        "frac": round(ach / PEAK_FP32_MFMA_TFLOPS, 4), "traffic": traffic,
                        # x + f + y (+ residual) of the kernel's launches, every operand once, per launch: what the
                        # PMC `traffic` (same unit) is to be compared with
                        "algorithmic_bytes": int(d["bytes"] / max(1, d["launches"])),
                        "traffic_source": tsrc, "traffic_profile_commit": tcommit,
                        "mode": "serial pass (one stream): launch durations of isolated kernels, the kernel choices of the timed region",
                        "launches": d["launches"], "avg_launch_ms": round(d["ms"] / d["launches"], 4),
                        "flop_per_launch": d["flops"] / d["launches"],
                        "all_conv_kernels": {"achieved": round(conv_fl / (conv_ms * 1e-3) / 1e12, 2),
                                             "ms_per_step": round(conv_ms / rsteps, 3)},
                        "steps": rsteps,
                        "per_kernel": [{"kernel": r["kernel"], "ms_per_step": round(r["ms"] / rsteps, 3),
                                        "tflops": round(r["flops"] / (r["ms"] * 1e-3) / 1e12, 2),
                                        "launches_per_step": r["launches"] // rsteps} for r in rows[:8]]}

    # ---- CPU baseline leg (rank 0, N = 1): the oracle on a bounded sample ------------------
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline and F == 1:
        cpu = cpu_baseline(wl, args.cpu_pairs, W)

    # ---- north_star configuration (bounded second pass, child process: its own nets and tuning lookups) --------
    north = None
    want_ns = args.north_star == 1 or (args.north_star < 0 and wl == "distill" and world == 1 and F == 1 and
                                       not args.per_gpu_batch and args.teacher == "resnet50" and not args.serial
                                       and W == 300 and not args.teacher_batch)
    if rank == 0 and world == 1 and want_ns:
        north = north_star_pass()

    # student FLOPs scale with the spectrogram width (16.633 GFLOP at W = 300, 22.33 at the reference's default W = 400)
    student_gflop = 22.33 if W == 400 else GFLOP["student_fwd_bwd_300"] * (W / 300.0)
    if wl == "distill":
        gflop_unit = (win["frames"] / nb if win is not None else F) * GFLOP["%s_fwd" % args.teacher] + student_gflop
    elif wl == "student":
        gflop_unit = student_gflop
    elif wl == "teacher":
        gflop_unit = GFLOP["senet50_fwd"]
    else:
        gflop_unit = GFLOP["senet50_fwd_bwd"] + student_gflop

    # proof of N RCCL ranks: ncclCommCount of the communicator the exchange runs on (the library's through
    # xm_comm_count, or torch's nccl group); null when the exchange does not run over RCCL (gloo debug runs, N = 1
    # without XM_DEBUG_DIST).  `world` is the launcher's process count -- a different fact.
    rccl_ranks = parserv.rccl_count()
    if rank == 0:
        out = {
            "metric": "distillation-step samples/sec (face+audio pair)" if wl in ("distill", "joint")
                      else ("student fwd+bwd samples/sec" if wl == "student" else "teacher fwd images/sec"),
            "value": round(value, 2), "unit": "pairs/s" if wl in ("distill", "joint") else "samples/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": {"distill": "BASELINE config 4 shard: frozen %s teacher fwd + VGGVox student fwd/bwd/SGD" % args.teacher +
                                               ("" if F == 1 else (", %d frames/pair" % F if win is None else
                                                ", imdb windows: %d-%d frames/pair (%d frames for %d pairs)" % (
                                                    win["min"], win["max"], win["frames"], nb))),
                                    "student": "VGGVox student fwd+bwd+update (BASELINE config 2)",
                                    "teacher": "senet50-ferplus teacher fwd (BASELINE config 3)",
                                    "joint": "senet50 teacher fwd+bwd + VGGVox student fwd+bwd (BASELINE config 5 shard)"}[wl],
                       "step": "run_distillation.m:170-182: teacher logits -> soft-target CE (T=2) -> backward -> "
                               "ParameterServer sum -> SGD-momentum",
                       "per_gpu_batch": nb, "global_batch": units, "face": "224x224x3",
                       "teacher_batch": (win["frames"] if win is not None else nb * F) if (wl == "distill" and F > 1) else
                                        nb * (tmult if wl == "distill" else 1),
                       "spectrogram": "512x%dx1" % W, "parallelism": "dp%d" % world,
                       "weights": "random-init (seeded)", "parameter_server": args.parserv,
                       "streams": "serial" if args.serial else
                                  {"distill": ("teacher in %d sample-slice lanes, then the student step + wgrad side stream" %
                                               args.teacher_lanes) if F > 1 else
                                              "teacher one batch ahead on its own stream + wgrad side stream",
                                   "student": "wgrad side stream", "joint": "wgrad side stream",
                                   "teacher": "%d sample-slice lanes" % args.teacher_lanes}[wl]},
            "model_tflops_per_gpu": round(value / world * gflop_unit / 1e3, 2),
            "model_frac_of_fp32_mfma_peak": round(value / world * gflop_unit / 1e3 / PEAK_FP32_MFMA_TFLOPS, 4),
            "settle_steps": settle, "windows": windows, "rccl_ranks": rccl_ranks, "world": world,
            "control_group": ctl_backend if dist.is_initialized() else None, "exec_hint": exec_hint,
            "roofline": roofline, "cpu_baseline": cpu,
        }
        if north is not None:
            out["north_star_b256"] = north
    if os.environ.get("XM_TUNE_SAVE") and rank == 0:
        vl.tune_save()          # persist the tile choices measured in this run (tools/collect_profiles.sh)
    # tear the process group down BEFORE printing: RCCL writes its banner / teardown lines to stdout
    # and the contract line must be the last thing rank 0 prints
    if dist.is_initialized():
        torch.cuda.synchronize()
        parserv.stop()
        dist.destroy_process_group()
    try:
        C.CDLL(None).fflush(None)        # anything RCCL / HIP still hold in C stdio buffers goes out first
    except OSError:
        pass
    if rank == 0:
        if world > 1:
            time.sleep(1.0)              # let the other ranks finish their teardown output
        sys.stdout.flush()
        print(json.dumps(out), flush=True)


def north_star_pass(steps=12, warmup=3):
    """north_star: "SENet50-teacher + VGGVox-student distillation step at batch 256" on this one GPU, timed by the same
    code (a child `bench.py --teacher senet50 --per-gpu-batch 256`: `steps` timed steps after the usual warm-up /
    settle phase) so that it falls inside the driver's clock around this invocation."""
    import subprocess
    cmd = [sys.executable, os.path.abspath(__file__), "--teacher", "senet50", "--per-gpu-batch", "256", "--steps",
           str(steps), "--warmup", str(warmup), "--no-cpu-baseline", "--no-roofline", "--north-star", "0"]
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "XM_DEBUG_DIST", "XM_BENCH_MARKS"):
        env.pop(k, None)
    try:
        r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, timeout=300, env=env, check=True)
        d = json.loads(r.stdout.decode().strip().splitlines()[-1])
    except (subprocess.SubprocessError, ValueError, IndexError, OSError) as e:
        return {"error": "%s: %s" % (type(e).__name__, str(e)[:80])}
    return {"value": d["value"], "unit": d["unit"], "ms_per_step": d["ms_per_step"], "steps": d["steps"],
            "settle_steps": d["settle_steps"], "per_gpu_batch": 256, "teacher": "senet50-ferplus",
            "model_frac": d["model_frac_of_fp32_mfma_peak"], "model_tflops": d["model_tflops_per_gpu"],
            "note": "same step code, SE-ResNet50 teacher fwd + student fwd/bwd/SGD on 256 pairs, one GPU"}


def _cpu_teacher(name, n, heads=False):
    """oracle fp32 path (MatConvNet's CPU algorithm: im2row + SGEMM per image) over the oracle's own graph tables"""
    from oracle import graphs as G
    g = G.resnet50_teacher(se=name.startswith("senet"), heads=heads)
    P = G.perturb_bn(G.make_params(g, 100), g, 101)
    x = G.face_batch(n, 1)
    ins = {"data": x}
    if heads:
        ins["label"] = np.asfortranarray((np.arange(n) % 8 + 1).reshape(1, 1, 1, n).astype(np.float32))
    G.forward(g, {"data": x[..., :1]}, P, mode="test", acc64=False, keep=("prediction",))   # thread pool warm-up
    t0 = time.perf_counter()
    G.forward(g, ins, P, mode="test", acc64=False, keep=("prediction", "objective", "top1error"))
    return time.perf_counter() - t0


def _cpu_pass(wl, pairs, W):
    """one timed pass of the oracle's fp32 path over `pairs` units of the workload; returns (seconds, GFLOP)"""
    from oracle import oracle as O, graphs as G
    t_total, gflop = 0.0, 0.0
    if wl in ("distill", "teacher", "joint"):
        name = "resnet50-ferplus" if wl == "distill" else "senet50-ferplus"
        if wl != "joint":
            t_total += _cpu_teacher(name, pairs)
            gflop += pairs * (GFLOP["resnet50_fwd"] if wl == "distill" else GFLOP["senet50_fwd"])
        else:
            g = G.resnet50_teacher(se=True, heads=True)
            P = G.perturb_bn(G.make_params(g, 300), g, 301)
            x = G.face_batch(pairs, 5)
            lab = np.asfortranarray((np.arange(pairs) % 8 + 1).reshape(1, 1, 1, pairs).astype(np.float32))
            t0 = time.perf_counter()
            V = G.forward(g, {"data": x, "label": lab}, P, mode="normal", acc64=False)
            G.backward(g, V, {"objective": np.float32(1)}, P, mode="normal", acc64=False)
            t_total += time.perf_counter() - t0
            gflop += pairs * GFLOP["senet50_fwd_bwd"]
    if wl in ("distill", "student", "joint"):
        g = G.vggvox_student(W)
        P = G.make_params(g, 200)
        data, lgo, lab = G.spectrogram_batch(pairs, W, 2)
        t0 = time.perf_counter()
        V = G.forward(g, {"data": data, "logitTarget": lgo, "maxLabel": lab}, P, mode="normal", acc64=False)
        _, DP = G.backward(g, V, {"objective": np.float32(1)}, P, mode="normal", acc64=False)
        for k, d in DP.items():
            if not k.endswith("x"):
                O.sgd_update(P[k], np.zeros_like(P[k]), d.reshape(P[k].shape, order="F"), 1e-4, 0.9, 5e-4, pairs)
        t_total += time.perf_counter() - t0
        gflop += pairs * GFLOP["student_fwd_bwd_300"] * (W / 300.0)
    return t_total, gflop


def imdb_windows(nb, W, seed):
    """ragged logit windows of `nb` pairs as getBatchEmoVoxCeleb.m:137-158 draws them: the teacher's logits exist for every 6th
    frame of a 25 fps track (time2idx, :210-214); a training sample is a random window of audSamp samples (:67-68) of the
    track's audio, and its target aggregates the rows time2idx(start) .. min(time2idx(end), rows) (:145-158).  Synthetic imdb:
    track lengths uniform in 4-20 s (VoxCeleb utterances are at least 4 s), every 10th track short of up to 3 rows (the
    'ffmpeg did not produce the exact number of frames' case the reference clamps, :150-156).  Returns 1-based inclusive row
    ranges into the concatenation of the windows' frames -- the teacher runs on exactly the frames the windows use."""
    from mcncrossmodalemotions_amd import batch as xbatch
    rng = np.random.default_rng(seed + 77)
    fs = 16000.0
    aud = xbatch.aud_samples(W)
    first, last, pos = [], [], 1
    for i in range(nb):
        dur = float(rng.uniform(4.0, 20.0))
        rows = xbatch.time2idx(dur)
        if i % 10 == 9:
            rows = max(1, rows - int(rng.integers(1, 4)))
        wr = int(rng.integers(1, max(2, int(dur * fs - aud))))                  # randi(numel(z) - audSamp)
        a = xbatch.time2idx(wr / fs)
        b = min(xbatch.time2idx((wr + aud - 1) / fs), rows)
        a = min(a, b)
        n = b - a + 1
        first.append(pos)
        last.append(pos + n - 1)
        pos += n
    cnt = [b - a + 1 for a, b in zip(first, last)]
    return {"first": first, "last": last, "frames": pos - 1, "min": min(cnt), "max": max(cnt)}


def cpu_baseline(wl, pairs, W, passes=3, budget_s=10.0):
    """MatConvNet-CPU-equivalent restatement (oracle fp32 path: im2row + vectorised SGEMM, images in parallel,
    OpenMP, one thread pinned to each physical core by oracle.pin_threads -- no more threads than the cgroup's CPU
    quota pays for) on a bounded sample of the same workload: up to `passes` timed passes inside ~`budget_s` seconds
    of CPU work (always at least one), the MEDIAN is the value, min / max, the quota and the host's load average are
    in the line (a shared host moved single passes by 2.4 x between boxes in round 4: other tenants' load).  Checker
    code used as a timed baseline only; a "port", NOT MatConvNet itself: a tuned BLAS behind the real vl_nnconv
    would be faster still (this SGEMM reaches a fraction of the cores' peak), so read the GPU / CPU ratio as an
    upper bound."""
    from oracle import oracle as O
    quota = O.cpu_quota()
    try:
        load1 = os.getloadavg()[0]
    except OSError:
        load1 = None
    cores, restore = O.pin_threads()     # one thread per physical core this process may use, pinned to it
    if pairs <= 0:
        pairs = max(8, cores // 16)      # ~3 s per pass on an idle 128-core host
    runs = []
    try:
        t_all = time.perf_counter()
        for _ in range(max(1, passes)):
            runs.append(_cpu_pass(wl, pairs, W))
            spent = time.perf_counter() - t_all
            if spent + spent / len(runs) > budget_s:      # the next pass would not fit
                break
    finally:
        restore()
    ts = sorted(t for t, _ in runs)
    med, gflop = ts[len(ts) // 2], runs[0][1]
    return {"value": round(pairs / med, 4), "unit": "pairs/s" if wl in ("distill", "joint") else "samples/s",
            "cores": cores, "kind": "port", "cpu": host_info(), "gflops": round(gflop / med, 1),
            "passes": len(ts), "min": round(pairs / ts[-1], 4), "max": round(pairs / ts[0], 4),
            "quota": None if quota is None else round(quota, 2), "loadavg_1m": None if load1 is None else round(load1, 1),
            "sample": "median of %d pass(es) over %d unit(s) of the same workload (%.1f s in all), oracle fp32 path: "
                      "im2row + vectorised SGEMM, OpenMP x %d pinned to physical cores (min(affinity, cgroup quota)); "
                      "a restatement, not MatConvNet" % (len(ts), pairs, sum(ts), cores)}


def cpu_teacher_line(args):
    """BASELINE config 1 as its own line (teacher/benchmark_ferplus_models.m:46-54): resnet50-ferplus forward with
    the loss / classerror heads attached, test mode, batch 32, on the host cores only -- no GPU is touched.  The
    arithmetic is the oracle's fp32 path (MatConvNet's CPU algorithm shape); median of the timed passes."""
    import torch  # noqa: F401  (first, as everywhere: its OpenMP runtime is then the one the oracle shares)
    from oracle import oracle as O
    O.set_num_threads()
    nb = args.per_gpu_batch or 32
    K, Wm = max(1, args.steps or 3), max(1, args.warmup or 1)
    for _ in range(Wm):
        _cpu_teacher("resnet50-ferplus", nb, heads=True)
    ts = [_cpu_teacher("resnet50-ferplus", nb, heads=True) for _ in range(K)]
    med = float(np.median(ts))
    out = {"metric": "teacher fwd images/sec (MatConvNet-CPU-equivalent restatement, host cores)",
           "value": round(nb / med, 3), "unit": "samples/s", "n_gpus": 0, "steps": K, "warmup": Wm,
           "ms_per_step": round(med * 1e3, 2), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
           "dtype": "f32", "data": "synthetic",
           "config": {"workload": "resnet50-ferplus teacher forward + softmaxlog / classerror heads, test mode "
                                  "(BASELINE config 1; benchmark_ferplus_models.m:46-54)",
                      "per_gpu_batch": nb, "global_batch": nb, "face": "224x224x3", "parallelism": "cpu",
                      "weights": "random-init (seeded)"},
           "cpu_baseline": {"value": round(nb / med, 3), "unit": "samples/s", "cores": O.num_threads(), "kind": "port",
                            "cpu": host_info(), "gflops": round(nb * GFLOP["resnet50_fwd"] / med, 1),
                            "sample": "%d passes of %d images, median; min %.2f s max %.2f s" % (K, nb, min(ts), max(ts))},
           "roofline": None}
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
