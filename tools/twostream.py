"""Experiment: frozen-teacher forward as two half-batches on two HIP streams vs one full batch."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mcncrossmodalemotions_amd import vl, zoo, batch as xbatch

dev = torch.device("cuda", 0)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 32
NAME = sys.argv[2] if len(sys.argv) > 2 else "resnet50-ferplus"
NS = int(sys.argv[3]) if len(sys.argv) > 3 else 2
def mk():
    t = zoo.ferPlusZoo(NAME, seed=100); zoo.strip_losses(t); t.move("gpu")
    t.vars["prediction"].precious = True
    zoo.calibrate_moments(t, ["data", xbatch.getImageBatch(16, seed=9, device=dev)]); t.mode = "test"
    return t
nets = [mk() for _ in range(NS)]
ta = nets[0]
faces = xbatch.getImageBatch(N, seed=4, device=dev)
base = faces.permute(3, 2, 1, 0).contiguous()
parts = [base[i * (N // NS):(i + 1) * (N // NS)].permute(3, 2, 1, 0) for i in range(NS)]
full = base.permute(3, 2, 1, 0)
streams = [torch.cuda.Stream() for _ in range(NS)]

def one():
    ta.eval(["data", full])
def two():
    cur = torch.cuda.current_stream()
    for st, net, part in zip(streams, nets, parts):
        st.wait_stream(cur)
        with torch.cuda.stream(st):
            net.eval(["data", part])
    for st in streams:
        cur.wait_stream(st)

def timeit(fn, reps=30):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3
print("%s one stream, batch %d: %.3f ms" % (NAME, N, timeit(one)))
print("%d streams x %d : %.3f ms" % (NS, N // NS, timeit(two)))
