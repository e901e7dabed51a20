"""Seeded synthetic k-mer models and reads.

The 9-mer models of the reference are absent from its checkout
(/root/reference/.MISSING_LARGE_BLOBS), and pod5/bam inputs cannot be produced in
this image, so every configuration of BASELINE.json runs on data from these
generators. Only the *generator* is committed, never the 12 MB model file.

Model file format = the reference's TSV (models/README.md, aligner.cpp:88-143):
header ``kmer\\tlevel_mean\\tlevel_stdv`` then one line per k-mer, lexicographic
ACGT order, k-mers written 5'->3'.
"""
from __future__ import annotations

import os
from dataclasses import dataclass

import numpy as np

BASES = "ACGT"
PORES = {
    # pore string -> (enum value, rna, k)   (aligner_bindings.cpp:18-32, aligner.cpp:62-86)
    "rna002": (0, True, 5),
    "rna004": (1, True, 9),
    "dna_r9": (2, False, 5),
    "dna_r10_260bps": (3, False, 9),
    "dna_r10_400bps": (4, False, 9),
}


def kmer_strings(k: int) -> list[str]:
    """All 4**k k-mers in lexicographic ACGT order."""
    n = 4 ** k
    codes = np.arange(n, dtype=np.int64)
    digits = np.empty((n, k), dtype=np.int64)
    for i in range(k):
        digits[:, k - 1 - i] = codes % 4
        codes //= 4
    lut = np.frombuffer(BASES.encode(), dtype="S1")
    return ["".join(x) for x in lut[digits].astype("U1")]


def model_values(k: int, seed: int = 7, stdev: float = 0.15) -> tuple[np.ndarray, np.ndarray]:
    """(mean, stdev) per k-mer in FILE order (lexicographic 5'->3')."""
    rng = np.random.default_rng(seed)
    mean = rng.standard_normal(4 ** k)
    sd = np.full(4 ** k, stdev)
    return mean, sd


def write_model(path: str, k: int, seed: int = 7, stdev: float = 0.15) -> str:
    """Write a synthetic model TSV; values use repr() so they round-trip exactly."""
    mean, sd = model_values(k, seed, stdev)
    return write_model_values(path, k, mean, sd)


def write_model_values(path: str, k: int, mean: np.ndarray, sd: np.ndarray) -> str:
    """Model TSV from explicit per-k-mer values in FILE order (lexicographic 5'->3')."""
    names = kmer_strings(k)
    tmp = path + ".tmp%d" % os.getpid()
    with open(tmp, "w") as w:
        w.write("kmer\tlevel_mean\tlevel_stdv\n")
        w.write("".join(f"{names[i]}\t{float(mean[i])!r}\t{float(sd[i])!r}\n" for i in range(len(names))))
    os.replace(tmp, path)
    return path


def code_order_table(mean_file: np.ndarray, sd_file: np.ndarray, k: int, rna: bool):
    """Re-index file-order values into the aligner's k-mer-code order.

    The loader stores file k-mer ``s`` at code(reverse(s)) for RNA pores and at
    code(s) for DNA (aligner.cpp:136-141).
    """
    n = 4 ** k
    if not rna:
        return mean_file.copy(), sd_file.copy()
    codes = np.arange(n, dtype=np.int64)
    rev = np.zeros(n, dtype=np.int64)
    c = codes.copy()
    for _ in range(k):
        rev = rev * 4 + c % 4
        c //= 4
    mean = np.empty(n)
    sd = np.empty(n)
    mean[rev] = mean_file
    sd[rev] = sd_file
    return mean, sd


def read_model_file(path: str):
    """Parse a model TSV -> (kmers list, mean array, stdev array) in file order."""
    kmers, mean, sd = [], [], []
    with open(path) as f:
        next(f)
        for line in f:
            p = line.rstrip("\n").split("\t")
            kmers.append(p[0])
            mean.append(float(p[1]))
            sd.append(float(p[2]))
    return kmers, np.array(mean), np.array(sd)


@dataclass
class SynthRead:
    signal: np.ndarray  # float64, normalised, aligner orientation
    sequence: str       # aligner orientation (RNA: 3'->5', starts with polyA)


def _seq_codes(digits: np.ndarray, k: int) -> np.ndarray:
    kc = len(digits) - k + 1
    code = np.zeros(kc, dtype=np.int64)
    for j in range(k):
        code = code * 4 + digits[j:j + kc]
    return code


def make_read(rng: np.random.Generator, mean_code: np.ndarray, sd_code: np.ndarray, k: int,
              n_bases: int, dwell: float, rna: bool, polya=None) -> SynthRead:
    """One read: i.i.d. bases, per-k-mer dwell max(2, Poisson(dwell)), samples
    N(mean_k, (c*stdev_k)^2) with c ~ U(0.8, 2.0). Guarantees S >= 2*Kc. ``polya`` = (lo, hi): the read goes on with that
    many more A's behind the pad, the way a real direct-RNA read starts with its tail."""
    digits = rng.integers(0, 4, size=n_bases)
    if rna:
        digits[:9] = 0  # aligner-orientation RNA reads start with the polyA pad (segment.py:155-158)
    if polya is not None:
        run = min(int(rng.integers(polya[0], polya[1] + 1)), max(0, n_bases - 11))
        digits[:9 + run] = 0
        digits[9 + run] = int(rng.integers(1, 4))
    return read_from_digits(rng, digits, mean_code, sd_code, k, dwell)


def read_from_digits(rng: np.random.Generator, digits: np.ndarray, mean_code: np.ndarray, sd_code: np.ndarray, k: int,
                     dwell: float) -> SynthRead:
    """The signal half of make_read for a prescribed base sequence (0..3 = ACGT, aligner orientation)."""
    codes = _seq_codes(digits, k)
    dw = np.maximum(2, rng.poisson(dwell, size=len(codes)))
    c = rng.uniform(0.8, 2.0)
    idx = np.repeat(codes, dw)
    sig = mean_code[idx] + c * sd_code[idx] * rng.standard_normal(len(idx))
    seq = "".join(BASES[d] for d in digits)
    return SynthRead(np.ascontiguousarray(sig, dtype=np.float64), seq)


def make_reads(seed: int, n_reads: int, pore: str, mean_file: np.ndarray, sd_file: np.ndarray,
               n_bases, dwell: float | None = None, polya=None) -> list[SynthRead]:
    """``n_bases`` is an int or a (lo, hi) range sampled uniformly per read."""
    _, rna, k = PORES[pore]
    if dwell is None:
        dwell = 10.0 if rna else 12.5
    mean_code, sd_code = code_order_table(mean_file, sd_file, k, rna)
    rng = np.random.default_rng(seed)
    out = []
    for _ in range(n_reads):
        nb = n_bases if isinstance(n_bases, int) else int(rng.integers(n_bases[0], n_bases[1] + 1))
        out.append(make_read(rng, mean_code, sd_code, k, nb, dwell, rna, polya))
    return out


# BASELINE.json configs
CONFIGS = {
    "cfg1": dict(pore="rna002", n_reads=1, n_bases=200, seed=1),
    "cfg2": dict(pore="rna004", n_reads=1024, n_bases=2000, seed=2),
    "cfg3": dict(pore="dna_r10_400bps", n_reads=4096, n_bases=(800, 8000), seed=3),
    "cfg4": dict(pore="rna004", n_reads=32768, n_bases=2000, seed=4),
    "cfg5": dict(pore="rna004", n_reads=8192, n_bases=2000, seed=5),
    # not a BASELINE config: cfg2 with reads that start with a polyA tail of 20-150 bases, as real direct-RNA reads do --
    # every read carries a structural tie (bench.py --workload cfg2_polya: what bit-exact borders cost on such data)
    "cfg2_polya": dict(pore="rna004", n_reads=1024, n_bases=2000, seed=2, polya=(20, 150)),
    # not a BASELINE config: reads SHORTER than the band is wide (N < 400 lattice columns: bw = N / 2 < 200,
    # NT_aligner_api.cpp:243) -- the narrow-band case, whose cells cost a lane slot each whether they are in the band or not
    "short_reads": dict(pore="rna004", n_reads=16384, n_bases=(150, 400), seed=6),
}


def pack_reads(reads: list[SynthRead]):
    """Flatten to the C-ABI batch layout: concatenated signals / sequences + offsets."""
    sig_off = np.zeros(len(reads) + 1, dtype=np.uint64)
    seq_off = np.zeros(len(reads) + 1, dtype=np.uint64)
    for i, r in enumerate(reads):
        sig_off[i + 1] = sig_off[i] + len(r.signal)
        seq_off[i + 1] = seq_off[i] + len(r.sequence)
    signals = np.concatenate([r.signal for r in reads]) if reads else np.zeros(0)
    seqs = "".join(r.sequence for r in reads).encode()
    return np.ascontiguousarray(signals, dtype=np.float64), sig_off, seqs, seq_off


def write_dataset(outdir: str, name: str, reads: list[SynthRead], pore: str, seed: int = 0,
                  drop_polya_every: int = 3, lead: int = 37, sm: float = 90.0, sd: float = 15.0,
                  container: str = "npz", pod5_chunk_samples: int = 102400, replicate: int = 1, basecalls: str = "tsv"):
    """Write reads as the vendor-free containers of ``dynamont_amd.pod5_io``: ``<name>.dynraw.npz``
    (int16 ADC + calibration) and ``<name>.dynbam.tsv`` (the BAM fields segment.py:222-256 reads).

    Reads are given in aligner orientation. For RNA pores the basecall is stored 5'->3' (reversed)
    and every ``drop_polya_every``-th read loses its polyA pad, which the harness must re-add
    (segment.py:155-158). ``lead`` untrimmed samples precede each signal (tag ts).
    Returns (raw_path, basecalls_path, expected) where expected[i] = (normalised float64 signal the
    harness must reconstruct, aligner-orientation sequence).
    ``container="pod5"`` writes ``<name>.pod5`` (``dynamont_amd.pod5_native.write_pod5``, UUID read ids)
    instead of the .npz container; ``basecalls="bam"`` writes ``<name>.bam`` (``dynamont_amd.bam_io.write_bam``: an
    unaligned BAM with the tags dorado writes, sm/sd/qs single precision) instead of the TSV.
    ``replicate`` > 1 writes every read that many times under distinct read ids (the whole list, then again): a large
    dataset for throughput runs from a small number of generated reads. ``expected`` covers the first copy."""
    import os
    import uuid
    _, rna, _k = PORES[pore]
    raw_name = f"{name}.pod5" if container == "pod5" else f"{name}.dynraw.npz"
    rng = np.random.default_rng(seed)
    scale, offset = 0.1755, -240.0
    adcs, offs, ids, rows, expected, records = [], [0], [], [], [], []
    for i, r in enumerate(reads):
        pa = r.signal * sd + sm
        adc = np.rint(pa / scale - offset).astype(np.int16)
        pre = rng.integers(300, 900, size=lead).astype(np.int16)
        full = np.concatenate([pre, adc])
        rid = str(uuid.UUID(int=(seed << 64) | i)) if container == "pod5" else f"read-{seed}-{i:05d}"
        ids.append(rid)
        adcs.append(full)
        offs.append(offs[-1] + len(full))
        seq = r.sequence
        if rna:
            bam_seq = seq[::-1]
            if drop_polya_every and i % drop_polya_every == 1 and bam_seq.endswith("A" * 9):
                bam_seq = bam_seq[:-9]
        else:
            bam_seq = seq
        qs = float(np.round(rng.uniform(8.0, 20.0), 3))
        rows.append(f"{rid}\t{bam_seq}\t{qs}\t*\t{len(full)}\t{lead}\t*\t{raw_name}\t{sm}\t{sd}\n")
        records.append((rid, bam_seq, {"qs": qs, "ns": len(full), "ts": lead, "fn": raw_name, "sm": float(sm), "sd": float(sd)}))
        # what the harness reconstructs: float64((adc+offset)*scale in float32), -sm, /sd
        pa32 = (adc.astype(np.float32) + np.float32(offset)) * np.float32(scale)
        x = pa32.astype(np.float64)
        x -= sm
        x /= sd
        expected.append((x, seq if not rna else ("A" * 9 + bam_seq[::-1] if not bam_seq[::-1].startswith("A" * 9) else bam_seq[::-1])))
    n0 = len(reads)
    for rep in range(1, max(1, replicate)):
        for i in range(n0):
            j = rep * n0 + i
            rid = str(uuid.UUID(int=(seed << 64) | j)) if container == "pod5" else f"read-{seed}-{j:05d}"
            ids.append(rid)
            adcs.append(adcs[i])  # the same array object: write_pod5 compresses it once
            offs.append(offs[-1] + len(adcs[i]))
            rows.append(rid + rows[i][rows[i].index("\t"):])
            records.append((rid, records[i][1], records[i][2]))
    os.makedirs(outdir, exist_ok=True)
    raw = os.path.join(outdir, raw_name)
    if container == "pod5":
        from dynamont_amd.pod5_native import write_pod5
        write_pod5(raw, ids, adcs, np.full(len(ids), offset), np.full(len(ids), scale), chunk_samples=pod5_chunk_samples)
    else:
        np.savez(raw, read_ids=np.array(ids), offsets=np.array(offs, dtype=np.int64), adc=np.concatenate(adcs),
                 cal_scale=np.full(len(ids), scale), cal_offset=np.full(len(ids), offset))
    if basecalls == "bam":
        from dynamont_amd.bam_io import write_bam
        bam = os.path.join(outdir, f"{name}.bam")
        write_bam(bam, records)
    else:
        bam = os.path.join(outdir, f"{name}.dynbam.tsv")
        with open(bam, "w") as w:
            w.write("query_name\tsequence\tqs\tpi\tns\tts\tsp\tfn\tsm\tsd\n")
            w.writelines(rows)
    return raw, bam, expected
