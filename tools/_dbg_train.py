import sys, os, tempfile, subprocess
sys.path.insert(0, '/root/repo')
flags = sys.argv[1] if len(sys.argv) > 1 else ""
env = dict(os.environ, DYN_HIPCC_EXTRA=flags)
subprocess.run([sys.executable, "-c", "import dynamont_amd._native as n; n.build(force=True)"], env=env, check=True)
import numpy as np
from dynamont_amd import Aligner, synth
d = tempfile.mkdtemp()
model = synth.write_model(os.path.join(d, "syn9.model"), 9, seed=7, stdev=0.15)
_, mean, sd = synth.read_model_file(model)
cfg = synth.CONFIGS["cfg5"]
reads = synth.make_reads(cfg["seed"], 512, cfg["pore"], mean, sd, cfg["n_bases"])
al = Aligner(model, cfg["pore"], device=0)
res = al.train_batch([r.signal for r in reads], [r.sequence for r in reads])
bad = []
for i, r in enumerate(reads):
    a, n = int(res.em_offsets[i]), int(res.em_count[i])
    dlt = res.em_weight[a:a + n].sum() - len(r.signal)
    if abs(dlt) > 1e-6:
        bad.append((i, round(float(dlt), 6)))
print("flags [%s]: %d bad reads: %s" % (flags, len(bad), bad[:16]), flush=True)
