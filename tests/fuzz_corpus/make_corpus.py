#!/usr/bin/env python
"""Seeds of the host-side fuzz (tools/sanitize/fuzz_host.cpp), regenerated with `python tests/fuzz_corpus/make_corpus.py`:
  spec.bam    the BAM tests/test_format_pinning.py assembles from the SAM specification (BGZF blocks cut inside records
              and inside the header; every tag type the job generator reads, plus B arrays / A / H tags it steps over)
  chunk.vbz   one POD5 VBZ signal chunk (zstd around StreamVByte-16 of zigzag-coded deltas), chunk.i16 the samples it holds
  model5.tsv  a 5-mer model file in the reference's format (header line, kmer<TAB>mean<TAB>stdev)
Small on purpose: every mutation re-reads the whole seed."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from dynamont_amd import pod5_native, synth  # noqa: E402
import test_format_pinning as fp  # noqa: E402


def main():
    recs, _ = fp._records()
    data = fp.spec_bam(recs, refs=[("chr1", 1000000)])
    rng = np.random.default_rng(11)
    cuts = [int(x) for x in rng.integers(1, len(data), size=5)] + [7, 9]
    open(os.path.join(HERE, "spec.bam"), "wb").write(fp._blocks(data, cuts))
    x = np.cumsum(rng.integers(-40, 41, size=3000)).astype(np.int16)
    x[100] = 32767
    x[101] = -32768  # the largest deltas: three-byte codes never occur in svb16, two-byte ones at their limit
    open(os.path.join(HERE, "chunk.vbz"), "wb").write(pod5_native.vbz_compress(x))
    open(os.path.join(HERE, "chunk.i16"), "wb").write(x.astype("<i2").tobytes())
    synth.write_model(os.path.join(HERE, "model5.tsv"), 5, seed=7, stdev=0.25)
    for f in ("spec.bam", "chunk.vbz", "chunk.i16", "model5.tsv"):
        print(f, os.path.getsize(os.path.join(HERE, f)), "bytes")


if __name__ == "__main__":
    main()
