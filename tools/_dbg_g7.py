import sys, os, subprocess
sys.path.insert(0, '/root/repo')
env = dict(os.environ, DYN_HIPCC_EXTRA="-DDYN_DBG_LIN")
subprocess.run([sys.executable, "-c", "import dynamont_amd._native as n; n.build(force=True)"], env=env, check=True)
import numpy as np, tempfile
from dynamont_amd import Aligner, synth
g = np.load('/root/repo/tests/golden/g7_train.npz', allow_pickle=True)
d = tempfile.mkdtemp()
pore = str(g["t0_pore"])
k = 9 if "004" in pore else 5
sys.path.insert(0, '/root/repo/tests')
import conftest
print(pore, len(g["t0_signal"]))
model = synth.write_model(os.path.join(d, "m.model"), k, seed=7, stdev=0.25 if k == 5 else 0.15)
al = Aligner(model, pore, device=0)
res = al.train_batch([g["t0_signal"]], [str(g["t0_sequence"])])
reads = synth.make_reads(78, 3, pore, *synth.read_model_file(model)[1:], (200, 300))
r2 = al.train_batch([r.signal for r in reads], [r.sequence for r in reads])
print(res.status, res.Z, res.transitions[:3], g["t0_trans"])
