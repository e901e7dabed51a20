/* multi_gpu_example.c -- a plain C host driving several GPUs through ONE handle (dyn_multi_*).
 *
 *   gcc -O2 -Iinclude integration/multi_gpu_example.c -Ldynamont_amd -ldynamont_mi -Wl,-rpath,$PWD/dynamont_amd -lm -o multi_gpu_example
 *   ./multi_gpu_example model.tsv rna004 0 1 2 3          # device ordinals; an ordinal may repeat
 *
 * It builds a small synthetic batch from the model file itself (every read walks a random k-mer chain and emits
 * each k-mer's mean + noise), aligns it once on a single device and once across all listed devices, and checks that
 * both give the same segmentation. This is the whole contract a C/C++ consumer needs: no Python, no RCCL. */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "dynamont_mi.h"

static unsigned long long rng_state = 88172645463325252ull;
static double uniform01(void) {
  rng_state ^= rng_state << 13;
  rng_state ^= rng_state >> 7;
  rng_state ^= rng_state << 17;
  return (double)(rng_state >> 11) / 9007199254740992.0;
}
static double gauss(void) { return sqrt(-2.0 * log(uniform01() + 1e-300)) * cos(6.283185307179586 * uniform01()); }

static void* xcalloc(size_t n, size_t sz) {
  void* p = calloc(n ? n : 1, sz);
  if (!p) { fprintf(stderr, "out of memory\n"); exit(2); }
  return p;
}

static dyn_align_out make_out(uint64_t n, uint64_t cap) {
  dyn_align_out o;
  memset(&o, 0, sizeof o);
  o.Z = xcalloc(n, sizeof(double));
  o.status = xcalloc(n, sizeof(int32_t));
  o.bad_char = xcalloc(n, 1);
  o.seg_offsets = xcalloc(n + 1, sizeof(uint64_t));
  o.n_segments = xcalloc(n, sizeof(uint64_t));
  o.sequence_positions = xcalloc(cap, sizeof(uint64_t));
  o.signal_positions = xcalloc(cap, sizeof(uint64_t));
  o.probabilities = xcalloc(cap, sizeof(double));
  o.states = xcalloc(cap, 1);
  o.capacity = cap;
  return o;
}

int main(int argc, char** argv) {
  if (argc < 4) { fprintf(stderr, "usage: %s MODEL PORE DEVICE [DEVICE ...]\n", argv[0]); return 2; }
  char err[1024];
  int pore = 0, devs[16], nd = 0;
  if (dyn_pore_from_string(argv[2], &pore, err, sizeof err) != DYN_OK) { fprintf(stderr, "%s\n", err); return 2; }
  for (int i = 3; i < argc && nd < 16; ++i) devs[nd++] = atoi(argv[i]);

  dyn_multi* m = NULL;
  if (dyn_multi_create(argv[1], pore, "basic", 1, 400, devs, nd, &m, err, sizeof err) != DYN_OK) { fprintf(stderr, "%s\n", err); return 1; }
  dyn_aligner* a0 = dyn_multi_handle(m, 0);
  dyn_info info;
  dyn_aligner_info(a0, &info);
  const int k = info.kmer_size;
  double* model = xcalloc(2 * info.num_kmers, sizeof(double));
  dyn_aligner_model(a0, model);

  /* synthetic batch: n reads of 150..400 bases, ~10 samples per k-mer */
  const uint64_t n = 48;
  uint64_t* seq_off = xcalloc(n + 1, sizeof(uint64_t));
  uint64_t* sig_off = xcalloc(n + 1, sizeof(uint64_t));
  char* seqs = xcalloc(n * 400 + 1, 1);
  double* sig = xcalloc(n * 400 * 16, sizeof(double));
  for (uint64_t r = 0; r < n; ++r) {
    const int len = 150 + (int)(uniform01() * 250);
    char* s = seqs + seq_off[r];
    for (int i = 0; i < len; ++i) s[i] = (info.rna && i < 9) ? 'A' : "ACGT"[(int)(uniform01() * 4) & 3];
    uint64_t pos = sig_off[r];
    for (int i = 0; i + k <= len; ++i) {
      uint64_t code = 0;  /* k-mer code as the aligner forms it: base-4 digits, first base most significant */
      for (int j = 0; j < k; ++j) code = code * 4 + (uint64_t)(strchr("ACGT", s[i + j]) - "ACGT");
      const int dwell = 4 + (int)(uniform01() * 12);
      for (int d = 0; d < dwell; ++d) sig[pos++] = model[2 * code] + 1.2 * model[2 * code + 1] * gauss();
    }
    seq_off[r + 1] = seq_off[r] + (uint64_t)len;
    sig_off[r + 1] = pos;
  }
  const uint64_t cap = dyn_segment_capacity(a0, n, seq_off);

  dyn_align_out one = make_out(n, cap), all = make_out(n, cap);
  int rc = dyn_align_batch(a0, n, sig, sig_off, seqs, seq_off, 1, &one);                 /* one device */
  if (rc != DYN_OK) { fprintf(stderr, "dyn_align_batch: %s\n", dyn_aligner_last_error(a0)); return 1; }
  rc = dyn_multi_align_batch(m, n, sig, sig_off, seqs, seq_off, 1, &all);                 /* all devices */
  if (rc != DYN_OK) { fprintf(stderr, "dyn_multi_align_batch: %s\n", dyn_multi_last_error(m)); return 1; }

  uint64_t segments = 0, bad = 0;
  for (uint64_t r = 0; r < n; ++r) {
    if (one.status[r] != all.status[r] || one.n_segments[r] != all.n_segments[r] || one.seg_offsets[r] != all.seg_offsets[r]) { ++bad; continue; }
    for (uint64_t i = 0; i < one.n_segments[r]; ++i) {
      const uint64_t j = one.seg_offsets[r] + i;
      if (one.signal_positions[j] != all.signal_positions[j] || one.probabilities[j] != all.probabilities[j]) { ++bad; break; }
    }
    segments += one.n_segments[r];
  }
  printf("%llu reads, %llu segments on %d device handle(s): %s\n", (unsigned long long)n, (unsigned long long)segments,
         dyn_multi_device_count(m), bad ? "MISMATCH" : "identical to the single-device result");
  dyn_multi_destroy(m);
  return bad ? 1 : 0;
}
