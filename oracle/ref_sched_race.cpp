/*
 * ref_sched_race.cpp -- start / stop loop over the REFERENCE's own JobScheduler, compiled unchanged from where it lies
 * (test infrastructure only; oracle/Makefile target `_ref/sched_race`; the twin of radixhashjoin_amd/host/sched_stress.cpp).
 *
 * Why it exists: one GPU run of tests/test_gpu_reference_seam.py stopped for ever in a binary that holds this scheduler.
 * JobScheduler::stop (JobScheduler.cpp:140-146) writes `done = true` and broadcasts cond_nonempty WITHOUT queueLock, while a
 * worker tests `q.empty() && !done && !bar` under the lock (JobScheduler.cpp:29-31): a worker between that test and
 * pthread_cond_wait misses the only wake-up, and pthread_join in stop() never returns.  Behind the GPU seam the inner workers
 * of every query thread (MainScheduler.cpp:6-14) are idle from init to stop, which is this loop.
 *
 *     _ref/sched_race [cycles]     prints a line every 10000 cycles; "completed" if it gets through
 * Measured in the build container (8 cores): progress stops after 10^4 - 8 * 10^4 cycles in 3 of 3 runs, both threads in
 * futex wait (/proc/<pid>/task/<tid>/wchan).  Run it under `timeout`.
 */
#include <cstdio>
#include <cstdlib>

#include "JobScheduler.h"

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
