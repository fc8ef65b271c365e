/*
 * ref_gpu_seam.cpp -- link-time seam that puts librhj_hip.so behind the UNMODIFIED reference (test
 * infrastructure only; the twin of ref_tap.cpp).
 *
 * oracle/Makefile target `_ref/join_seam` compiles the reference's Result.cpp with
 * -DmultiRadixHashJoin=refMultiRadixHashJoin (the CPU implementation keeps a different symbol and is never
 * called) and every other reference file unchanged, from the sources where they lie.  This file supplies the
 * symbol the sole caller (Query.cpp:186) binds to and forwards it to rhj_join: the reference's own parser,
 * filters, intermediates, MainScheduler and CLI then drive the GPU join.  The body is INTEGRATION.md Option B.
 *
 * Second symbol bound here: JobScheduler::stop.  The reference's (JobScheduler.cpp:140-146) sets `done` and broadcasts
 * WITHOUT queueLock, so an idle worker between its `!done` test and pthread_cond_wait (JobScheduler.cpp:29-31) sleeps through
 * the only wake-up and stop() never returns -- and behind this seam the inner workers of every query thread are idle from
 * init to stop (oracle/ref_sched_race.cpp reproduces it with the reference's files alone).  The Makefile compiles
 * JobScheduler.cpp with -Dstop=ref_stop_racy (the racy body keeps another symbol and is never called) and the body below
 * is the two-line fix INTEGRATION.md gives the maintainer: publish `done` under the lock.  -DRHJ_SEAM_SCHED_LOOP builds
 * the start/stop loop of ref_sched_race.cpp over THIS stop() instead of the GPU seam (`_ref/sched_race_fixed`).
 */
#include <cstdio>
#include <cstdlib>

#define multiRadixHashJoin refMultiRadixHashJoin
#include "Result.h"
#undef multiRadixHashJoin

#include "../include/rhj.h"

static_assert(sizeof(tuple) == sizeof(rhj_tuple) && sizeof(key_tuple) == sizeof(rhj_pair), "layouts");

void JobScheduler::stop()
{
    pthread_mutex_lock(&queueLock);
    done = true;                                   /* under the lock: a worker has either seen it or is already waiting */
    pthread_cond_broadcast(&cond_nonempty);
    pthread_mutex_unlock(&queueLock);
    for (size_t i = 0; i < num_of_threads; i++) pthread_join(threads[i], nullptr);
}

#ifdef RHJ_SEAM_SCHED_LOOP
int main(int argc, char **argv)
{
    const long cycles = argc > 1 ? atol(argv[1]) : 1000000;
    for (long i = 0; i < cycles; i++) {
        JobScheduler js;
        js.init(4);
        js.stop();
        js.destroy();
        if (i % 10000 == 0) { printf("%ld\n", i); fflush(stdout); }
    }
    puts("completed");
    return 0;
}
#else

static thread_local rhj_ctx *tls_ctx = nullptr;   /* one context per query thread (MainScheduler.cpp:6-14) */

/* Itanium-ABI name of Result::multiRadixHashJoin(JobScheduler&, relation&, relation&) */
extern "C" void _ZN6Result18multiRadixHashJoinER12JobSchedulerR8relationS3_(Result *self, JobScheduler *js,
                                                                            relation *R, relation *S)
{
    (void)js;
    if (!tls_ctx && rhj_init(0, &tls_ctx) != RHJ_OK) { fprintf(stderr, "%s\n", rhj_last_error(nullptr)); exit(EXIT_FAILURE); }
    void *page = nullptr;
    uint64_t count = 0;
    int rc = rhj_join(tls_ctx, (const rhj_tuple *)R->tuples, R->num_tuples, (const rhj_tuple *)S->tuples, S->num_tuples,
                      nullptr, &page, &count);
    if (rc != RHJ_OK) { fprintf(stderr, "%s\n", rhj_last_error(tls_ctx)); exit(EXIT_FAILURE); }
    if (const char *log = getenv("RHJ_SEAM_LOG")) {             /* lets a test see that the GPU path really ran */
        FILE *f = fopen(log, "a");
        if (f) { fprintf(f, "%llu %llu %llu\n", (unsigned long long)R->num_tuples, (unsigned long long)S->num_tuples,
                         (unsigned long long)count); fclose(f); }
    }
    if (!page) return;                             /* head stays nullptr -> isEmpty() -> NULL (Query.cpp:188-191) */
    self->head = (bucket_info *)page;
    self->capacity = count;
    self->size = count;
}
#endif
