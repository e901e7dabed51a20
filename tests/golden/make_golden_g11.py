#!/usr/bin/env python
"""Generate golden fixture G11: what the reference's NTK aligner (`--mode resquiggle`, mode "ntk") answers.

Runs only in the authoring container. A 30-line probe (written to a temporary directory, never into the repository)
is compiled against the reference's own sources where they lie -- /root/reference/src/cpp/{aligner,NTK_aligner_api}.cpp,
plain g++, no build system -- and calls dynamont::NTKAligner::align / ::train (include/dynamont/NTK_aligner_api.hpp)
on a handful of reads. The fixture stores inputs and the exception texts. Observation pinned here: in this snapshot
every read that passes validateInput / sequenceToKmers fails with
"NTK alignment failed: alignment scores do not match" (NTK_aligner_api.cpp:911-917), for 5-mer and 9-mer pores alike,
and train() is the base class's "Training is not implemented for this aligner" (aligner.cpp:38-44). Callers put these
texts into `.errors` (segment.py:172-176), so they are observable output the MI355X build reproduces.

    python tests/golden/make_golden_g11.py
"""
import json
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from dynamont_amd import synth  # noqa: E402

REF = "/root/reference"
PROBE = r'''
#include <cstdio>
#include <fstream>
#include <string>
#include <vector>
#include "dynamont/NTK_aligner_api.hpp"
int main(int argc, char** argv) {
  dynamont::NTKAligner a(argv[1], static_cast<dynamont::PoreType>(atoi(argv[2])), 1, 400);
  std::ifstream f(argv[3]);
  std::string seq, line;
  while (std::getline(f, seq) && std::getline(f, line)) {
    std::vector<double> sig; size_t pos = 0;
    while (pos < line.size()) { size_t e; sig.push_back(std::stod(line.substr(pos), &e)); pos += e; while (pos < line.size() && line[pos] == ' ') ++pos; }
    try { auto r = a.align(sig.data(), sig.size(), seq, true); printf("align\tOK\n"); }
    catch (const std::exception& e) { printf("align\t%s\n", e.what()); }
    try { a.train(sig.data(), sig.size(), seq); printf("train\tOK\n"); } catch (const std::exception& e) { printf("train\t%s\n", e.what()); }
  }
}
'''


def main():
    d = tempfile.mkdtemp(prefix="g11_")
    open(os.path.join(d, "probe.cpp"), "w").write(PROBE)
    subprocess.run(["g++", "-std=c++17", "-O2", "-w", f"-I{REF}/include", os.path.join(d, "probe.cpp"), f"{REF}/src/cpp/aligner.cpp",
                    f"{REF}/src/cpp/NTK_aligner_api.cpp", "-o", os.path.join(d, "probe")], check=True)
    out = {"cases": []}
    for pore, model_key, k, sd, n_bases in (("rna002", "syn5", 5, 0.25, (20, 60)), ("dna_r9", "syn5", 5, 0.25, (20, 60)), ("rna004", "syn9", 9, 0.15, (12, 16))):
        model = synth.write_model(os.path.join(d, f"{model_key}.model"), k, seed=7, stdev=sd)
        _, mean, sdv = synth.read_model_file(model)
        reads = [(r.sequence, [float(x) for x in r.signal]) for r in synth.make_reads(9, 3, pore, mean, sdv, n_bases)]
        if k == 5:
            reads += [("ACGTNACGTA", [0.1] * 40), ("ACGTACGTAC", [0.1] * 4), ("ACG", [0.1] * 40)]
        path = os.path.join(d, f"reads_{pore}.txt")
        with open(path, "w") as w:
            for seq, sig in reads:
                w.write(seq + "\n" + " ".join(repr(x) for x in sig) + "\n")
        res = subprocess.run([os.path.join(d, "probe"), model, str(synth.PORES[pore][0]), path], check=True, capture_output=True, text=True).stdout.splitlines()
        for i, (seq, sig) in enumerate(reads):
            out["cases"].append(dict(pore=pore, model=model_key, sequence=seq, signal=sig,
                                     align=res[2 * i].split("\t", 1)[1], train=res[2 * i + 1].split("\t", 1)[1]))
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "g11_ntk_messages.json"), "w") as f:
        json.dump(out, f, indent=0)
    for c in out["cases"]:
        print(c["pore"], len(c["signal"]), len(c["sequence"]), "|", c["align"], "|", c["train"])


if __name__ == "__main__":
    main()
