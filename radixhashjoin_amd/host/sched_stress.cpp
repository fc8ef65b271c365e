// sched_stress.cpp -- start / stop stress of the host-side JobScheduler mirror (rhj_compat.cpp), no GPU needed.
//
// The reference's JobScheduler::stop (JobScheduler.cpp:140-146) sets `done` and broadcasts without holding queueLock, so a
// worker that has tested `!done` under the lock but not yet reached pthread_cond_wait sleeps through the only wake-up and
// stop() never returns: the same loop against the reference's scheduler (Option B) stops making progress
// after some ten thousand cycles on an 8-core host.  The mirror publishes `done` under the queue mutex; this program is
// the proof: `cycles` schedulers started and stopped at once (the idle-worker window), and every few cycles one that runs
// jobs, takes a barrier (the mirror's barrier also waits for jobs in flight), and is stopped with jobs still queued.
//
//     sched_stress [cycles=20000] [workers=4]        prints "ok <cycles> cycles <jobs> jobs", exit 0
#include <atomic>
#include <cstdio>
#include <cstdlib>

#include "rhj_compat.h"

static std::atomic<unsigned long> g_ran{0};

struct CountJob : Job {
    int run() override { g_ran.fetch_add(1); return 0; }
};

static void *worker(void *arg)
{
    static_cast<JobScheduler *>(arg)->threadWork(nullptr);
    return nullptr;
}

int main(int argc, char **argv)
{
    const long cycles = argc > 1 ? atol(argv[1]) : 20000;
    const size_t workers = argc > 2 ? (size_t)atol(argv[2]) : 4;
    unsigned long scheduled = 0;
    for (long i = 0; i < cycles; i++) {
        JobScheduler js;
        js.init(workers, worker);
        if (i % 8 == 0) {
            for (int j = 0; j < 16; j++) { js.schedule(new CountJob()); scheduled++; }
            js.barrier();
            if (g_ran.load() != scheduled) { fprintf(stderr, "barrier returned with %lu of %lu jobs run\n", g_ran.load(), scheduled); return 1; }
            for (int j = 0; j < 5; j++) { js.schedule(new CountJob()); scheduled++; }     // still queued when stop() is called
        }
        js.stop();
        js.destroy();
        if (g_ran.load() != scheduled) { fprintf(stderr, "stop returned with %lu of %lu jobs run\n", g_ran.load(), scheduled); return 1; }
    }
    printf("ok %ld cycles %lu jobs\n", cycles, scheduled);
    return 0;
}
