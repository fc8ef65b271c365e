/*
 * rhj_oracle.c -- CPU ORACLE (test infrastructure, see rhj_oracle.h).
 *
 * A plain-C, single-threaded restatement of the reference's radix hash join:
 *   8-bit LSB radix histogram per row range  ->  per-range prefix + row-index scatter
 *   ->  serial bucket-major merge-gather  ->  per-bucket chained hash build (smaller side,
 *   modulus next_prime(n)) / probe (larger side, full 64-bit compare)  ->  128 KiB result pages.
 * The pthread job queue of the reference (JobScheduler.cpp:67-146) only decides WHEN each job
 * body runs; every job writes to private memory, so running the same bodies in queue order on
 * one thread gives the same bytes.  `nranges` keeps the reference's NUM_OF_THREADS row split.
 */
#include "rhj_oracle.h"

#include <stdlib.h>
#include <string.h>

/* ---------------------------------------------------------------- auxFun.cpp */

/* auxFun.cpp:4-22: first prime strictly greater than x, except x<2 -> 2 and x==3 -> 5
 * (so 2 -> 5 as well: the "next odd" start value for even x is x+1 = 3, which is rejected
 * by the %3 test). Candidates are odd numbers, trial division by 3 and 6k+-1. */
size_t orc_next_prime(size_t x)
{
    if (x < 2) return 2;
    if (x == 3) return 5;
    size_t c = (x & 1) ? x + 2 : x + 1;
    for (;; c += 2) {
        if (c % 3 == 0) continue;
        int prime = 1;
        for (size_t d = 5; d * d <= c; d += 6) {
            if (c % d == 0 || c % (d + 2) == 0) { prime = 0; break; }
        }
        if (prime) return c;
    }
}

/* auxFun.cpp:24-26 */
size_t orc_pow2(size_t e) { return (size_t)1 << e; }

/* ---------------------------------------------------------------- partitioner */

/* JobScheduler.cpp:149-155 (HistogramJob::run): hist[payload & (nbins-1)]++ over [start,end). */
void orc_histogram_range(const orc_tuple *t, size_t start, size_t end, size_t nbins, size_t *hist)
{
    const size_t mask = nbins - 1;
    for (size_t i = start; i < end; i++) hist[t[i].payload & mask]++;
}

/* JobScheduler.cpp:162-177 (PartitionJob::run): exclusive prefix of the range's own histogram
 * into sum_hist, then scatter of the GLOBAL row index i to idx[cursor[bin]++]. */
void orc_partition_range(const orc_tuple *t, size_t start, size_t end, size_t nbins,
                         const size_t *hist, size_t *sum_hist, size_t *idx)
{
    const size_t mask = nbins - 1;
    size_t *cursor = (size_t *)malloc(nbins * sizeof(size_t));
    size_t run = 0;
    for (size_t b = 0; b < nbins; b++) { sum_hist[b] = run; cursor[b] = run; run += hist[b]; }
    for (size_t i = start; i < end; i++) idx[cursor[t[i].payload & mask]++] = i;
    free(cursor);
}

/* structs.cpp:146-161: range 0 gets n/T rows; ranges 1..(n%T) get one more (NOT the first n%T). */
void orc_split_ranges(size_t n, int nranges, size_t *start, size_t *end)
{
    size_t q = n / (size_t)nranges, r = n % (size_t)nranges;
    start[0] = 0; end[0] = q;
    for (int i = 1; i < nranges; i++) {
        start[i] = end[i - 1];
        end[i] = start[i] + q;
        if (r > 0) { end[i]++; r--; }
    }
}

/* structs.cpp:144-204 (relation_info::hash_relation) with multiHistogram (111-121) and
 * multiPartition (123-134): per-range histograms, global histogram = sum (168-173), per-range
 * index partitions, then the serial bucket-major / range-minor gather into R' (183-194). */
void orc_hash_relation(const orc_tuple *rel, size_t n, size_t nbins, int nranges,
                       orc_tuple *out, size_t *histogram)
{
    size_t *start = (size_t *)malloc(sizeof(size_t) * (size_t)nranges);
    size_t *end = (size_t *)malloc(sizeof(size_t) * (size_t)nranges);
    size_t **hists = (size_t **)malloc(sizeof(size_t *) * (size_t)nranges);
    size_t **sums = (size_t **)malloc(sizeof(size_t *) * (size_t)nranges);
    size_t **idx = (size_t **)malloc(sizeof(size_t *) * (size_t)nranges);
    orc_split_ranges(n, nranges, start, end);

    for (int i = 0; i < nranges; i++) {
        hists[i] = (size_t *)calloc(nbins, sizeof(size_t));
        orc_histogram_range(rel, start[i], end[i], nbins, hists[i]);
    }
    memset(histogram, 0, nbins * sizeof(size_t));
    for (int i = 0; i < nranges; i++)
        for (size_t b = 0; b < nbins; b++) histogram[b] += hists[i][b];

    for (int i = 0; i < nranges; i++) {
        sums[i] = (size_t *)calloc(nbins, sizeof(size_t));
        idx[i] = (size_t *)malloc(sizeof(size_t) * (end[i] - start[i] + 1));
        orc_partition_range(rel, start[i], end[i], nbins, hists[i], sums[i], idx[i]);
    }

    size_t pos = 0;
    for (size_t b = 0; b < nbins; b++)
        for (int i = 0; i < nranges; i++)
            for (size_t e = sums[i][b]; e < sums[i][b] + hists[i][b]; e++)
                out[pos++] = rel[idx[i][e]];

    for (int i = 0; i < nranges; i++) { free(hists[i]); free(sums[i]); free(idx[i]); }
    free(hists); free(sums); free(idx); free(start); free(end);
}

/* structs.cpp:86-109 (singleHistogram + singlePartition, never called by the reference but
 * the executable specification of R'): stable LSB-radix partition. */
void orc_single_partition(const orc_tuple *rel, size_t n, size_t nbins, orc_tuple *out, size_t *histogram)
{
    const size_t mask = nbins - 1;
    memset(histogram, 0, nbins * sizeof(size_t));
    for (size_t i = 0; i < n; i++) histogram[rel[i].payload & mask]++;
    size_t *cur = (size_t *)malloc(nbins * sizeof(size_t));
    size_t run = 0;
    for (size_t b = 0; b < nbins; b++) { cur[b] = run; run += histogram[b]; }
    for (size_t i = 0; i < n; i++) out[cur[rel[i].payload & mask]++] = rel[i];
    free(cur);
}

/* ---------------------------------------------------------------- result pages */

/* Result.cpp:10-14: capacity = (131072 - sizeof(header)) / 16 = 8191; size starts == capacity so
 * the first add allocates; head == NULL means "empty result". */
void orc_result_init(orc_result *r)
{
    r->capacity = (ORC_PAGE_BYTES - sizeof(orc_page)) / sizeof(orc_tuple);
    r->size = r->capacity;
    r->head = NULL;
}

int orc_result_is_empty(const orc_result *r) { return r->head == NULL; }   /* Result.cpp:16-18 */

/* Result.cpp:21-35: LIFO page list, head = newest page, only head may be partially filled. */
void orc_result_add(orc_result *r, uint64_t k1, uint64_t k2)
{
    if (r->size == r->capacity) {
        orc_page *p = (orc_page *)malloc(ORC_PAGE_BYTES);
        p->next = r->head;
        r->head = p;
        r->size = 0;
    }
    orc_pair *slots = (orc_pair *)(r->head + 1);
    slots[r->size].keyR = k1;
    slots[r->size].keyS = k2;
    r->size++;
}

/* Result.cpp:78-84 */
void orc_result_add_all(orc_result *r, const orc_page *node, size_t n)
{
    const orc_pair *slots = (const orc_pair *)(node + 1);
    for (size_t i = 0; i < n; i++) orc_result_add(r, slots[i].keyR, slots[i].keyS);
}

/* Result.cpp:127-133 */
void orc_result_free(orc_result *r)
{
    while (r->head) { orc_page *p = r->head; r->head = p->next; free(p); }
}

/* ---------------------------------------------------------------- bucket join */

/* Result.cpp:43-76 (Result::join_buckets): chained hash table over the small side,
 * bucket[payload % p] = newest index, chain[k] = previous; probe walks the chain and emits on
 * full 64-bit equality.  order_flag != 0 means "big side is R" (JobScheduler.cpp:187-190), so
 * the emitted pair is always (rowID of R, rowID of S). */
void orc_join_buckets(orc_result *res, const orc_tuple *small_, const orc_tuple *big,
                      size_t beg_small, size_t beg_big, size_t small_size, size_t big_size, int order_flag)
{
    const size_t p = orc_next_prime(small_size);
    int64_t *bucket = (int64_t *)malloc(sizeof(int64_t) * p);
    int64_t *chain = (int64_t *)malloc(sizeof(int64_t) * (small_size ? small_size : 1));
    for (size_t i = 0; i < p; i++) bucket[i] = -1;

    for (size_t k = 0; k < small_size; k++) {
        size_t h = (size_t)(small_[beg_small + k].payload % p);
        chain[k] = bucket[h];
        bucket[h] = (int64_t)k;
    }
    for (size_t i = beg_big; i < beg_big + big_size; i++) {
        const uint64_t v = big[i].payload;
        for (int64_t k = bucket[v % p]; k != -1; k = chain[k]) {
            const orc_tuple *s = &small_[beg_small + (size_t)k];
            if (v == s->payload) {
                if (order_flag) orc_result_add(res, big[i].key, s->key);
                else            orc_result_add(res, s->key, big[i].key);
            }
        }
    }
    free(bucket); free(chain);
}

/* Result.cpp:90-124 (Result::multiRadixHashJoin) + JoinJob::run (JobScheduler.cpp:186-192):
 * partition both inputs on the 8 LSBs, one bucket join per bucket with both sides non-empty
 * (build on S when |R_b| >= |S_b|, else on R), then concatenate the per-bucket page lists in
 * bucket order into `out` by re-adding pair by pair (111-121). */
void orc_multi_radix_hash_join(const orc_tuple *R, size_t nR, const orc_tuple *S, size_t nS,
                               int nranges, orc_result *out)
{
    const size_t nb = orc_pow2(ORC_HASH_LSB);
    orc_tuple *Rp = (orc_tuple *)malloc(sizeof(orc_tuple) * (nR ? nR : 1));
    orc_tuple *Sp = (orc_tuple *)malloc(sizeof(orc_tuple) * (nS ? nS : 1));
    size_t *hR = (size_t *)malloc(sizeof(size_t) * nb);
    size_t *hS = (size_t *)malloc(sizeof(size_t) * nb);
    orc_hash_relation(R, nR, nb, nranges, Rp, hR);
    orc_hash_relation(S, nS, nb, nranges, Sp, hS);

    orc_result *res = (orc_result *)malloc(sizeof(orc_result) * nb);
    for (size_t b = 0; b < nb; b++) orc_result_init(&res[b]);

    size_t begR = 0, begS = 0;
    for (size_t b = 0; b < nb; b++) {
        if (hR[b] != 0 && hS[b] != 0) {
            if (hR[b] >= hS[b]) orc_join_buckets(&res[b], Sp, Rp, begS, begR, hS[b], hR[b], 1);
            else                orc_join_buckets(&res[b], Rp, Sp, begR, begS, hR[b], hS[b], 0);
        }
        begR += hR[b];
        begS += hS[b];
    }
    for (size_t b = 0; b < nb; b++) {
        if (orc_result_is_empty(&res[b])) continue;
        const orc_page *node = res[b].head;
        orc_result_add_all(out, node, res[b].size);
        for (node = node->next; node; node = node->next) orc_result_add_all(out, node, res[b].capacity);
        orc_result_free(&res[b]);
    }
    free(res); free(Rp); free(Sp); free(hR); free(hS);
}

/* ---------------------------------------------------------------- test helpers */

uint64_t orc_mix(uint64_t z)
{
    z += 0x9E3779B97F4A7C15ULL;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}

size_t orc_result_count(const orc_result *r)
{
    size_t n = 0, sz = r->size;
    for (const orc_page *p = r->head; p; p = p->next) { n += sz; sz = r->capacity; }
    return n;
}

static uint64_t pair_term(uint64_t kr, uint64_t ks) { return orc_mix(kr * 0x100000001B3ULL ^ orc_mix(ks)); }

uint64_t orc_result_checksum(const orc_result *r)
{
    uint64_t c = 0; size_t sz = r->size;
    for (const orc_page *p = r->head; p; p = p->next) {
        const orc_pair *s = (const orc_pair *)(p + 1);
        for (size_t i = 0; i < sz; i++) c += pair_term(s[i].keyR, s[i].keyS);
        sz = r->capacity;
    }
    return c;
}

uint64_t orc_pairs_checksum(const orc_pair *p, size_t n)
{
    uint64_t c = 0;
    for (size_t i = 0; i < n; i++) c += pair_term(p[i].keyR, p[i].keyS);
    return c;
}

void orc_result_flatten(const orc_result *r, orc_pair *out)
{
    size_t sz = r->size, pos = 0;
    for (const orc_page *p = r->head; p; p = p->next) {
        memcpy(out + pos, p + 1, sz * sizeof(orc_pair));
        pos += sz; sz = r->capacity;
    }
}

size_t orc_join_flat(const orc_tuple *R, size_t nR, const orc_tuple *S, size_t nS, orc_pair **out_pairs)
{
    orc_result res; orc_result_init(&res);
    orc_multi_radix_hash_join(R, nR, S, nS, ORC_NUM_OF_THREADS, &res);
    size_t n = orc_result_count(&res);
    *out_pairs = (orc_pair *)malloc(sizeof(orc_pair) * (n ? n : 1));
    orc_result_flatten(&res, *out_pairs);
    orc_result_free(&res);
    return n;
}

size_t orc_join_count_checksum(const orc_tuple *R, size_t nR, const orc_tuple *S, size_t nS, uint64_t *checksum)
{
    orc_result res; orc_result_init(&res);
    orc_multi_radix_hash_join(R, nR, S, nS, ORC_NUM_OF_THREADS, &res);
    size_t n = orc_result_count(&res);
    *checksum = orc_result_checksum(&res);
    orc_result_free(&res);
    return n;
}

void orc_gen_R(orc_tuple *R, size_t nR, uint64_t D)
{
    for (size_t i = 0; i < nR; i++) { R[i].key = i; R[i].payload = orc_mix(1 + (uint64_t)i % D); }
}

void orc_gen_S_chain(orc_tuple *S, size_t nS, uint64_t D)
{
    uint64_t s = 42;
    for (size_t j = 0; j < nS; j++) { s = orc_mix(s); S[j].key = j; S[j].payload = orc_mix(1 + s % D); }
}

void orc_gen_S_disjoint(orc_tuple *S, size_t nS, uint64_t D)
{
    for (size_t j = 0; j < nS; j++) { S[j].key = j; S[j].payload = orc_mix(D + 1 + (uint64_t)j); }
}

void orc_gen_const(orc_tuple *T, size_t n, uint64_t value)
{
    for (size_t i = 0; i < n; i++) { T[i].key = i; T[i].payload = value; }
}

void orc_gen_S_counter(orc_tuple *S, size_t nS, uint64_t D, uint64_t seed)
{
    for (size_t j = 0; j < nS; j++) { S[j].key = j; S[j].payload = orc_mix(1 + orc_mix((uint64_t)j ^ seed) % D); }
}
