import sys, os, importlib.util, numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
R = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
from oracle import graphs as G
from mcncrossmodalemotions_amd import vl, zoo
spec = importlib.util.spec_from_file_location("m", R + "/tests/golden/make_golden_nets.py"); M = importlib.util.module_from_spec(spec); spec.loader.exec_module(M)
Z = np.load(R + "/tests/golden/nets_full.npz")
net = zoo.emoVoxZoo("emovoxceleb-student", scratch=1, lossType="hot-cross-ent", numSeconds=3)
_, P = M.student_params()
for k, v in P.items(): net.params[k].value = np.asfortranarray(v)
net.pack_params()
net.fuse = os.environ.get("FUSE", "1") == "1"
data, lgo, lab = G.spectrogram_batch(M.STUDENT_N, M.STUDENT_W, M.STUDENT_IN_SEED)
net.vars["prediction"].precious = True
net.mode = "normal"
net.eval(["data", vl.from_numpy(data), "logitTarget", vl.from_numpy(lgo), "maxLabel", vl.from_numpy(lab)], ["objective", 1])
torch.cuda.synchronize()
print("pred err", np.abs(vl.to_numpy(net.vars["prediction"].value) - Z["stu_prediction"]).max(), "cpu32 dev", Z["stu_prediction_dev32"])
for name in net.params:
    flat = vl.to_numpy(net.params[name].der).ravel(order="F")
    idx = M.sample_idx(flat.size)
    ref = Z["stu_der_%s_samp" % name]
    err = np.abs(flat[idx] - ref).max()
    d32 = float(Z["stu_der_%s_dev32" % name])
    print("%-8s max|ref| %.3e  hip err %.3e  cpu32 dev %.3e  ratio %.2f  rel %.2e" % (name, np.abs(ref).max(), err, d32, err / max(d32, 1e-30), err/np.abs(ref).max()))
