/* TEST INFRASTRUCTURE ONLY -- see nt_oracle.c. */
#ifndef NT_ORACLE_H
#define NT_ORACLE_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct nto_model nto_model;

/* pore: 0 RNA002, 1 RNA004, 2 DNA_R9, 3 DNA_R10_260, 4 DNA_R10_400 */
nto_model* nto_model_load(const char* path, int pore, uint64_t band, char* err, uint64_t errcap);
void nto_model_free(nto_model* m);
uint64_t nto_model_num_kmers(const nto_model* m);
int nto_model_kmer_size(const nto_model* m);
/* copies (mean, stdev) interleaved, k-mer-code order */
void nto_model_table(const nto_model* m, double* out2n);

/* k-mer coding of a sequence; returns count or -1 (err set). */
int64_t nto_sequence_to_kmers(const nto_model* m, const char* seq, int32_t* out, char* err,
                              uint64_t errcap);

/* band window of row t: start (signed), nStart, nEnd. */
void nto_compute_bounds(uint64_t T, uint64_t N, uint64_t bandwidth, int64_t* start,
                        uint64_t* nStart, uint64_t* nEnd);

/* Returns number of segments (>=0) or -1 (err set). */
int64_t nto_align(const nto_model* m, const double* sig, uint64_t S, const char* seq, int calc,
                  double* Z, uint64_t* seqpos, uint64_t* sigpos, double* prob, char* state,
                  char* err, uint64_t errcap);

/* emission: 2*numKmers doubles or NULL. Also returns the raw sufficient statistics if
 * stats3n != NULL (weight, sum, sumSq per k-mer, 3*numKmers) and log-space transition
 * sums in logsums2 (newM1, newE2) if non-NULL. Returns numKmers or -1. */
int64_t nto_train(const nto_model* m, const double* sig, uint64_t S, const char* seq, double* Z,
                  double* trans3, double* emission, double* stats3n, double* logsums2, char* err,
                  uint64_t errcap);

/* Debug helper for kernel bring-up: dense dump of forward E / backward E on the T x N lattice
 * restricted to the band; cells outside the band are -inf. out arrays are T*W with
 * W = 2*bandwidth+1, entry (t, n - start_t). */
int64_t nto_debug_fb(const nto_model* m, const double* sig, uint64_t S, const char* seq,
                     double* fE, double* bE, double* fM, double* bM, uint64_t cap, char* err,
                     uint64_t errcap);

/* smallest |vM - vE| over the on-path traceback decisions of the last nto_align(calc=1) */
double nto_last_decision_margin(void);
/* the same over decisions between columns with different k-mers only (see nt_oracle.c) */
double nto_last_decision_margin_distinct(void);
/* row / column of the smallest margin and the two values compared there */
void nto_last_decision_margin_at(uint64_t* t, uint64_t* n, double* vm, double* ve);

double nto_log_normal_pdf(double x, double mean, double stdev);
double nto_log_plus(double x, double y);

#ifdef __cplusplus
}
#endif
#endif
