import sys, os
sys.path.insert(0, "/root/repo")
import numpy as np
from dynamont_amd import Aligner, synth
import tempfile
d = tempfile.mkdtemp()
model = synth.write_model(os.path.join(d, "m9.model"), 9, seed=7, stdev=0.15)
_, mean, sd = synth.read_model_file(model)
reads = synth.make_reads(1, 700, "rna004", mean, sd, (200, 400))
packed = synth.pack_reads(reads)
al = Aligner(model, "rna004", device=0)
for mode in [(True, 0), (True, 8), (False, 0), (True, 8)]:
    al.set_session_mode(*mode)
    s0 = al.session_stats()
    ts = [al.align_async(*packed, True) for _ in range(3)]
    for t in ts:
        t.wait(); tm = t.timing(); t.close()
    s1 = al.session_stats()
    print(mode, "launches", tm["launches"], "sessions", s1["sessions"] - s0["sessions"], "waves", s1["waves"] - s0["waves"], "err", al.last_error())
al.close()
