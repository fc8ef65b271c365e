/*
 * ref_shim.cpp -- C-ABI driver around the REAL reference (test infrastructure only).
 *
 * Our own code: it only #includes the reference's headers (found through -I$(REF), the sources
 * stay where they lie under /root/reference) and calls its public entry points.  oracle/Makefile
 * links it with the reference's Result.cpp / JobScheduler.cpp / structs.cpp / auxFun.cpp into
 * oracle/_ref/libref_rhj.so.  Used to (a) pin oracle/rhj_oracle.c, (b) generate tests/golden/,
 * (c) time the reference's pthread CPU path as bench.py's cpu_baseline ("kind": "reference").
 */
#include <chrono>
#include <cstdlib>
#include <cstring>

#include "Result.h"
#include "JobScheduler.h"
#include "auxFun.h"
#include "structs.h"

/* external-linkage dead code of the reference (structs.cpp:86-109): the serial partition spec */
size_t *singleHistogram(relation &rel, size_t twoInLSB);
tuple *singlePartition(relation &rel, size_t twoInLSB, const size_t *histogram);

/* JobScheduler::stop with `done` published under queueLock.  The reference's body (JobScheduler.cpp:140-146) sets it and
 * broadcasts without the lock, so a worker between its `!done` test and pthread_cond_wait (JobScheduler.cpp:29-31) sleeps
 * through the only wake-up and stop() never returns (oracle/ref_sched_race.cpp); every entry point below stops its scheduler
 * right after the last job, which is where that window is.  The Makefile compiles JobScheduler.cpp with
 * -Dstop=ref_stop_racy, so the racy body keeps another symbol and is never called.  Same body as in ref_gpu_seam.cpp. */
void JobScheduler::stop()
{
    pthread_mutex_lock(&queueLock);
    done = true;
    pthread_cond_broadcast(&cond_nonempty);
    pthread_mutex_unlock(&queueLock);
    for (size_t i = 0; i < num_of_threads; i++) pthread_join(threads[i], nullptr);
}

static void fill(relation &dst, const void *src, size_t n)
{
    dst.num_tuples = n;
    dst.tuples = new tuple[n ? n : 1];          /* relation::~relation does delete[] (structs.cpp:210-212) */
    if (n) memcpy(dst.tuples, src, n * sizeof(tuple));
}

extern "C" {

int ref_num_threads(void) { return NUM_OF_THREADS; }

size_t ref_next_prime(size_t x) { return next_prime(x); }

/* Runs Result::multiRadixHashJoin (Result.cpp:90) on copies of R,S.  *out_pairs = malloc'd
 * {rowR,rowS} array in page order, *head_null = Result::isEmpty(), *seconds = wall time of the
 * multiRadixHashJoin call alone (scheduler start-up and input copies excluded). Returns count. */
size_t ref_join(const void *R, size_t nR, const void *S, size_t nS,
                void **out_pairs, int *head_null, double *seconds)
{
    relation relR, relS;
    fill(relR, R, nR);
    fill(relS, S, nS);
    JobScheduler js;
    js.init(NUM_OF_THREADS);
    size_t count = 0;
    {
        Result res;
        auto t0 = std::chrono::steady_clock::now();
        res.multiRadixHashJoin(js, relR, relS);
        auto t1 = std::chrono::steady_clock::now();
        if (seconds) *seconds = std::chrono::duration<double>(t1 - t0).count();
        if (head_null) *head_null = res.isEmpty() ? 1 : 0;
        size_t sz = res.size;
        for (bucket_info *n = res.head; n; n = n->next) { count += sz; sz = res.capacity; }
        if (out_pairs) {
            key_tuple *flat = (key_tuple *)malloc(sizeof(key_tuple) * (count ? count : 1));
            size_t pos = 0;
            sz = res.size;
            for (bucket_info *n = res.head; n; n = n->next) {
                memcpy(flat + pos, &n[1], sz * sizeof(key_tuple));
                pos += sz; sz = res.capacity;
            }
            *out_pairs = flat;
        }
    }
    js.stop();
    js.destroy();
    return count;
}

/* relation_info::hash_relation (structs.cpp:144): out[n] = R', hist[nbins] */
void ref_hash_relation(const void *rel, size_t n, size_t nbins, void *out, size_t *hist)
{
    relation r;
    fill(r, rel, n);
    JobScheduler js;
    js.init(NUM_OF_THREADS);
    {
        relation_info info;
        info.hash_relation(js, r, nbins);
        if (n) memcpy(out, info.tuples.tuples, n * sizeof(tuple));
        memcpy(hist, info.histogram, nbins * sizeof(size_t));
    }
    js.stop();
    js.destroy();
}

/* singleHistogram + singlePartition (structs.cpp:86-109) */
void ref_single_partition(const void *rel, size_t n, size_t nbins, void *out, size_t *hist)
{
    relation r;
    fill(r, rel, n);
    size_t *h = singleHistogram(r, nbins);
    tuple *t = singlePartition(r, nbins, h);
    if (n) memcpy(out, t, n * sizeof(tuple));
    memcpy(hist, h, nbins * sizeof(size_t));
    delete[] h;
    delete[] t;
}

void ref_free(void *p) { free(p); }

}  /* extern "C" */
