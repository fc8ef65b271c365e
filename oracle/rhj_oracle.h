/*
 * rhj_oracle.h -- CPU ORACLE for the radix-hash-join hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * This is a plain-C restatement of the algorithm of pelekoudasq/radixHashJoin's
 * Result::multiRadixHashJoin path (reference file:line cited per function).  It exists
 * to CHECK the HIP product path (tests/, __graft_entry__.smoke(), bench.py's cpu_baseline
 * leg).  Nothing under radixhashjoin_amd/ may include, link, import or execute it.
 *
 * Parity status: PINNED.  The restatement is checked (tests/test_oracle.py) against
 *   (1) the known answers the real reference produced (tests/golden/),
 *   (2) the reference itself compiled from /root/reference into oracle/_ref/ (when present).
 */
#ifndef RHJ_ORACLE_H
#define RHJ_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* reference structs.h:33-36 -- NB: `key` is the rowID, `payload` is the join value. */
typedef struct { uint64_t key; uint64_t payload; } orc_tuple;
/* reference Result.h:9-12 */
typedef struct { uint64_t keyR; uint64_t keyS; } orc_pair;

/* reference Result.h:14-17 (page header; pairs follow the header in the same malloc block) */
typedef struct orc_page { struct orc_page *next; } orc_page;
/* reference Result.h:19-38 (data members only) */
typedef struct { size_t capacity; size_t size; orc_page *head; } orc_result;

#define ORC_NUM_OF_THREADS 8          /* reference JobScheduler.h:11 */
#define ORC_HASH_LSB 8                /* reference Result.cpp:5       */
#define ORC_PAGE_BYTES (128 * 1024)   /* reference Result.cpp:7       */

size_t orc_next_prime(size_t x);                               /* auxFun.cpp:4-22  */
size_t orc_pow2(size_t e);                                     /* auxFun.cpp:24-26 */

/* HistogramJob::run, JobScheduler.cpp:149-155 */
void orc_histogram_range(const orc_tuple *t, size_t start, size_t end, size_t nbins, size_t *hist);
/* PartitionJob::run, JobScheduler.cpp:162-177: writes GLOBAL row indices, bucket-major, into idx[0..end-start) */
void orc_partition_range(const orc_tuple *t, size_t start, size_t end, size_t nbins,
                         const size_t *hist, size_t *sum_hist, size_t *idx);
/* range split of relation_info::hash_relation, structs.cpp:146-161 */
void orc_split_ranges(size_t n, int nranges, size_t *start, size_t *end);
/* relation_info::hash_relation, structs.cpp:144-204.  out[n] = R', histogram[nbins]. */
void orc_hash_relation(const orc_tuple *rel, size_t n, size_t nbins, int nranges,
                       orc_tuple *out, size_t *histogram);
/* dead-code serial spec singleHistogram+singlePartition, structs.cpp:86-109 */
void orc_single_partition(const orc_tuple *rel, size_t n, size_t nbins, orc_tuple *out, size_t *histogram);

void orc_result_init(orc_result *r);                           /* Result.cpp:10-14   */
int  orc_result_is_empty(const orc_result *r);                 /* Result.cpp:16-18   */
void orc_result_add(orc_result *r, uint64_t k1, uint64_t k2);  /* Result.cpp:21-35   */
void orc_result_add_all(orc_result *r, const orc_page *node, size_t n); /* Result.cpp:78-84 */
void orc_result_free(orc_result *r);                           /* Result.cpp:127-133 */
/* Result::join_buckets, Result.cpp:43-76 */
void orc_join_buckets(orc_result *res, const orc_tuple *small_, const orc_tuple *big,
                      size_t beg_small, size_t beg_big, size_t small_size, size_t big_size, int order_flag);
/* Result::multiRadixHashJoin, Result.cpp:90-124 (+ JoinJob::run, JobScheduler.cpp:186-192) */
void orc_multi_radix_hash_join(const orc_tuple *R, size_t nR, const orc_tuple *S, size_t nS,
                               int nranges, orc_result *out);

/* -------- helpers for tests (not part of the reference) -------- */
uint64_t orc_mix(uint64_t z);                                  /* SURVEY.md §8d splitmix64 step */
size_t   orc_result_count(const orc_result *r);
/* order-insensitive checksum of SURVEY.md App. A: sum of mix(keyR*0x100000001B3 ^ mix(keyS)) */
uint64_t orc_result_checksum(const orc_result *r);
uint64_t orc_pairs_checksum(const orc_pair *p, size_t n);
/* copy all pairs (page order) into out[count] */
void     orc_result_flatten(const orc_result *r, orc_pair *out);
/* one-call convenience: join, write malloc'd pair array; returns count */
size_t   orc_join_flat(const orc_tuple *R, size_t nR, const orc_tuple *S, size_t nS, orc_pair **out_pairs);
size_t   orc_join_count_checksum(const orc_tuple *R, size_t nR, const orc_tuple *S, size_t nS, uint64_t *checksum);

/* synthetic generators of SURVEY.md App. A.  D = value-domain size (D == nR for PK/FK). */
void orc_gen_R(orc_tuple *R, size_t nR, uint64_t D);           /* R[i] = {i, mix(1 + i % D)}                   */
void orc_gen_S_chain(orc_tuple *S, size_t nS, uint64_t D);     /* s=42; s=mix(s); S[j] = {j, mix(1 + s % D)}   */
void orc_gen_S_disjoint(orc_tuple *S, size_t nS, uint64_t D);  /* S[j] = {j, mix(D + 1 + j)}                   */
void orc_gen_const(orc_tuple *T, size_t n, uint64_t value);    /* T[i] = {i, value}                            */
/* counter-based variant used by the device generator (radixhashjoin_amd gen v2): S[j] = {j, mix(1 + mix(j ^ seed) % D)} */
void orc_gen_S_counter(orc_tuple *S, size_t nS, uint64_t D, uint64_t seed);

#ifdef __cplusplus
}
#endif
#endif
