// rhj_compat.cpp -- implementation of the reference-surface mirror (rhj_compat.h) over the C-ABI.
#include "rhj_compat.h"

#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <queue>
#include <thread>

// job queue + host threads of a scheduler started with init(n, start_routine): the reference's
// queue / mutex / condvars / pthread_t[] (JobScheduler.h:88-96) with std::thread instead of pthreads.
struct WorkerPool {
    std::mutex mu;
    std::condition_variable cv_work, cv_idle;
    std::queue<Job *> q;
    std::vector<std::thread> threads;
    size_t busy = 0;
    bool done = false;
};

namespace {

[[noreturn]] void die(rhj_ctx *ctx, const char *what, int rc)
{
    // the reference's failure style: message on stderr, exit(EXIT_FAILURE) (JobScheduler.cpp:24-27)
    fprintf(stderr, "rhj: %s failed (%d): %s\n", what, rc, rhj_last_error(ctx));
    exit(EXIT_FAILURE);
}

int log2_exact(size_t v)
{
    int b = 0;
    while (((size_t)1 << b) < v) b++;
    if (((size_t)1 << b) != v) { fprintf(stderr, "rhj: twoInLSB must be a power of two\n"); exit(EXIT_FAILURE); }
    return b;
}

constexpr size_t PAGE_BYTES = 128 * 1024;     // BUCKET_SIZE, Result.cpp:7

struct DevMem {                               // scoped HBM allocation through the C-ABI
    rhj_ctx *ctx;
    void *p = nullptr;
    DevMem(rhj_ctx *c, uint64_t bytes) : ctx(c)
    {
        int rc = rhj_dev_alloc(ctx, bytes ? bytes : 16, &p);
        if (rc != RHJ_OK) die(ctx, "rhj_dev_alloc", rc);
    }
    ~DevMem() { rhj_dev_free(ctx, p); }
};

}  // namespace

// ---------------------------------------------------------------------------------------- scheduler
void Job::init(void *) {}

bool JobScheduler::init(size_t n)
{
    num_of_threads = n;
    const char *dev = getenv("RHJ_DEVICE");
    int rc = rhj_init(dev ? atoi(dev) : 0, &ctx_);
    if (rc != RHJ_OK) die(nullptr, "rhj_init", rc);
    return true;
}

// JobScheduler.cpp:67-82.  The start routine receives `this`, exactly like the pthread start routines of the
// reference (jobThreadWork / mainThreadWork); it is expected to end up in threadWork().
bool JobScheduler::init(size_t n, void *start_routine(void *))
{
    num_of_threads = n;
    pool_ = new WorkerPool();
    for (size_t i = 0; i < n; i++) pool_->threads.emplace_back([this, start_routine]() { (void)start_routine(this); });
    return true;
}

// JobScheduler.cpp:22-64: run queued jobs until stop() has been called and the queue is drained.
void JobScheduler::threadWork(void *arg)
{
    if (!pool_) return;                               // no queue: init(n) schedulers run jobs on the caller
    WorkerPool &p = *pool_;
    for (;;) {
        Job *job = nullptr;
        {
            std::unique_lock<std::mutex> lk(p.mu);
            p.cv_work.wait(lk, [&] { return p.done || !p.q.empty(); });
            if (p.q.empty()) return;                  // done and drained
            job = p.q.front();
            p.q.pop();
            p.busy++;
        }
        job->gpu = ctx_;
        job->init(arg);                               // QueryJob::init receives the query thread's private JobScheduler
        job->run();
        delete job;                                   // JobScheduler.cpp:53
        {
            std::lock_guard<std::mutex> lk(p.mu);
            p.busy--;
            if (p.q.empty() && p.busy == 0) p.cv_idle.notify_all();
        }
    }
}

int JobScheduler::schedule(Job *job)
{
    if (pool_) {
        std::lock_guard<std::mutex> lk(pool_->mu);
        pool_->q.push(job);
        pool_->cv_work.notify_one();
        return 0;
    }
    job->gpu = ctx_;
    job->init(job_arg_);
    job->run();
    delete job;                               // the reference's worker deletes the job (JobScheduler.cpp:53)
    return 0;
}

void JobScheduler::barrier()
{
    if (pool_) {
        std::unique_lock<std::mutex> lk(pool_->mu);
        pool_->cv_idle.wait(lk, [&] { return pool_->q.empty() && pool_->busy == 0; });
        return;
    }
    if (!ctx_) return;
    int rc = rhj_sync(ctx_);
    if (rc != RHJ_OK) die(ctx_, "rhj_sync", rc);
}

void JobScheduler::stop()
{
    if (pool_) {                                      // finish every queued job, then join the threads
        {
            std::lock_guard<std::mutex> lk(pool_->mu);
            pool_->done = true;
        }
        pool_->cv_work.notify_all();
        for (std::thread &t : pool_->threads) t.join();
        pool_->threads.clear();
        return;
    }
    barrier();
}

bool JobScheduler::destroy()
{
    if (pool_) { delete pool_; pool_ = nullptr; }
    if (ctx_) rhj_destroy(ctx_);
    ctx_ = nullptr;
    return true;
}

// ---------------------------------------------------------------------------------------- jobs
HistogramJob::HistogramJob(size_t *histogram, relation &rel, size_t twoInLSB, size_t start, size_t end)
    : histogram(histogram), rel(rel), twoInLSB(twoInLSB), start(start), end(end) {}

// hist[payload & (twoInLSB-1)]++ over rows [start,end), added into `histogram` (JobScheduler.cpp:149-155)
int HistogramJob::run()
{
    const size_t n = end - start;
    const int bits = log2_exact(twoInLSB);
    DevMem d_rel(gpu, n * sizeof(tuple)), d_hist(gpu, twoInLSB * 8);
    int rc = rhj_copy_h2d(gpu, d_rel.p, rel.tuples + start, n * sizeof(tuple));
    if (rc == RHJ_OK) rc = rhj_histogram(gpu, (const rhj_tuple *)d_rel.p, n, 0, bits, (uint64_t *)d_hist.p);
    uint64_t *h = new uint64_t[twoInLSB];
    if (rc == RHJ_OK) rc = rhj_copy_d2h(gpu, h, d_hist.p, twoInLSB * 8);
    if (rc != RHJ_OK) die(gpu, "HistogramJob", rc);
    for (size_t b = 0; b < twoInLSB; b++) histogram[b] += (size_t)h[b];
    delete[] h;
    return 0;
}

PartitionJob::PartitionJob(size_t *tuples, relation &rel, size_t twoInLSB, size_t start, size_t end,
                           size_t *sumHistogram, size_t *histogram)
    : tuples(tuples), rel(rel), twoInLSB(twoInLSB), start(start), end(end), sumHistogram(sumHistogram),
      histogram(histogram) {}

// Row indices of [start,end) grouped by bucket into `tuples`, sumHistogram = exclusive prefix of the
// range's histogram (JobScheduler.cpp:162-177).  Order inside a bucket is unspecified here (the
// reference's is ascending; nothing downstream observes it).
int PartitionJob::run()
{
    const size_t n = end - start;
    const int bits = log2_exact(twoInLSB);
    // the partition kernel moves whole tuples: give it {key = global row index, payload = join value}
    tuple *tmp = new tuple[n ? n : 1];
    for (size_t i = 0; i < n; i++) { tmp[i].key = start + i; tmp[i].payload = rel.tuples[start + i].payload; }
    DevMem d_in(gpu, n * sizeof(tuple)), d_out(gpu, n * sizeof(tuple)), d_ps(gpu, (twoInLSB + 1) * 8);
    int rc = rhj_copy_h2d(gpu, d_in.p, tmp, n * sizeof(tuple));
    if (rc == RHJ_OK) rc = rhj_partition(gpu, (const rhj_tuple *)d_in.p, n, bits, 0, (rhj_tuple *)d_out.p, (uint64_t *)d_ps.p);
    if (rc == RHJ_OK) rc = rhj_copy_d2h(gpu, tmp, d_out.p, n * sizeof(tuple));
    uint64_t *ps = new uint64_t[twoInLSB + 1];
    if (rc == RHJ_OK) rc = rhj_copy_d2h(gpu, ps, d_ps.p, (twoInLSB + 1) * 8);
    if (rc != RHJ_OK) die(gpu, "PartitionJob", rc);
    for (size_t i = 0; i < n; i++) tuples[i] = (size_t)tmp[i].key;
    for (size_t b = 0; b < twoInLSB; b++) sumHistogram[b] = (size_t)ps[b];
    (void)histogram;                          // input of the reference's prefix; the device recomputes it
    delete[] ps;
    delete[] tmp;
    return 0;
}

JoinJob::JoinJob(Result &result, relation_info *relShashed, relation_info *relRhashed, size_t begS, size_t begR,
                 size_t histS, size_t histR)
    : result(result), relShashed(relShashed), relRhashed(relRhashed), begS(begS), begR(begR), histS(histS),
      histR(histR) {}

// build on the smaller bucket, keep (rowR,rowS) order (JobScheduler.cpp:186-192)
int JoinJob::run()
{
    if (histR >= histS)
        result.join_slices(gpu, relShashed->tuples.tuples + begS, histS, relRhashed->tuples.tuples + begR, histR, true);
    else
        result.join_slices(gpu, relRhashed->tuples.tuples + begR, histR, relShashed->tuples.tuples + begS, histS, false);
    return 0;
}

// ---------------------------------------------------------------------------------------- relations
relation::~relation() { delete[] tuples; }

relation_info::~relation_info() { delete[] histogram; }

// R' = rel grouped by (payload & (twoInLSB-1)), histogram[twoInLSB] (structs.cpp:144-204): one
// rhj_partition call instead of 8 HistogramJobs + 8 PartitionJobs + the serial merge-gather.
void relation_info::hash_relation(JobScheduler &js, relation &rel, size_t twoInLSB)
{
    rhj_ctx *ctx = js.context();
    const uint64_t n = rel.num_tuples;
    const int bits = log2_exact(twoInLSB);
    tuples.num_tuples = n;
    tuples.tuples = new tuple[n ? n : 1];
    histogram = new size_t[twoInLSB]();
    DevMem d_in(ctx, n * sizeof(tuple)), d_out(ctx, n * sizeof(tuple)), d_ps(ctx, (twoInLSB + 1) * 8);
    int rc = rhj_copy_h2d(ctx, d_in.p, rel.tuples, n * sizeof(tuple));
    if (rc == RHJ_OK) rc = rhj_partition(ctx, (const rhj_tuple *)d_in.p, n, bits, 0, (rhj_tuple *)d_out.p, (uint64_t *)d_ps.p);
    if (rc == RHJ_OK) rc = rhj_copy_d2h(ctx, tuples.tuples, d_out.p, n * sizeof(tuple));
    uint64_t *ps = new uint64_t[twoInLSB + 1];
    if (rc == RHJ_OK) rc = rhj_copy_d2h(ctx, ps, d_ps.p, (twoInLSB + 1) * 8);
    if (rc != RHJ_OK) die(ctx, "hash_relation", rc);
    for (size_t b = 0; b < twoInLSB; b++) histogram[b] = (size_t)(ps[b + 1] - ps[b]);
    delete[] ps;
}

// ---------------------------------------------------------------------------------------- Result
Result::Result()
{
    capacity = (PAGE_BYTES - sizeof(bucket_info)) / sizeof(key_tuple);      // 8191, Result.cpp:10-14
    size = capacity;
    head = nullptr;
}

Result::~Result()
{
    while (head) { bucket_info *p = head; head = head->next; free(p); }     // Result.cpp:127-133
}

bool Result::isEmpty() { return head == nullptr; }

void Result::add_result(uint64_t keyR, uint64_t keyS)
{
    if (size == capacity) {                                                  // Result.cpp:21-35
        // pages always hold `capacity` pairs: 128 KiB with the default capacity, larger after a
        // multiRadixHashJoin installed one device-filled page
        size_t bytes = sizeof(bucket_info) + capacity * sizeof(key_tuple);
        if (bytes < PAGE_BYTES) bytes = PAGE_BYTES;
        bucket_info *page = (bucket_info *)malloc(bytes);
        page->next = head;
        head = page;
        size = 0;
    }
    key_tuple *slots = (key_tuple *)&head[1];
    slots[size].keyR = keyR;
    slots[size].keyS = keyS;
    size++;
}

void Result::addAll(bucket_info *node, size_t n)
{
    const key_tuple *slots = (const key_tuple *)&node[1];                   // Result.cpp:78-84
    for (size_t i = 0; i < n; i++) add_result(slots[i].keyR, slots[i].keyS);
}

// One bucket pair through the device: small side builds, big side probes; pairs appended in
// (rowR,rowS) order (Result.cpp:43-76).  orderFlag: the big side is R.
void Result::join_slices(rhj_ctx *ctx, const tuple *small, size_t nSmall, const tuple *big, size_t nBig, bool orderFlag)
{
    const tuple *R = orderFlag ? big : small, *S = orderFlag ? small : big;
    const size_t nR = orderFlag ? nBig : nSmall, nS = orderFlag ? nSmall : nBig;
    void *page = nullptr;
    uint64_t count = 0;
    int rc = rhj_join(ctx, (const rhj_tuple *)R, nR, (const rhj_tuple *)S, nS, nullptr, &page, &count);
    if (rc != RHJ_OK) die(ctx, "rhj_join", rc);
    if (page) {
        addAll((bucket_info *)page, count);
        free(page);
    }
}

void Result::join_buckets(relation_info *small, relation_info *big, size_t begSmall, size_t begBig, size_t histSmall,
                          size_t histBig, bool orderFlag)
{
    // no scheduler in this signature (Result.h:32): use a short-lived context
    rhj_ctx *ctx = nullptr;
    const char *dev = getenv("RHJ_DEVICE");
    int rc = rhj_init(dev ? atoi(dev) : 0, &ctx);
    if (rc != RHJ_OK) die(nullptr, "rhj_init", rc);
    join_slices(ctx, small->tuples.tuples + begSmall, histSmall, big->tuples.tuples + begBig, histBig, orderFlag);
    rhj_destroy(ctx);
}

// The drop-in (Result.h:30, Result.cpp:90-124).  `this` must be freshly constructed, as at its only
// call site (Query.cpp:185-186).  Afterwards: head == nullptr when nothing matched, else ONE page
// with capacity == size == number of pairs and next == nullptr -- every consumer reads capacity/size
// at run time (intermediate.cpp:151-179), so the invariant "head has `size` pairs, every other page
// `capacity`" holds.
void Result::multiRadixHashJoinBatch(JobScheduler &js, size_t n, relation *const *R, relation *const *S, Result *const *results)
{
    std::vector<rhj_join_desc> d(n);
    std::vector<void *> pages(n, nullptr);
    std::vector<uint64_t> counts(n, 0);
    for (size_t i = 0; i < n; i++)
        d[i] = rhj_join_desc{(const rhj_tuple *)R[i]->tuples, R[i]->num_tuples, (const rhj_tuple *)S[i]->tuples, S[i]->num_tuples};
    const int rc = rhj_join_batch(js.context(), (uint32_t)n, d.data(), pages.data(), counts.data());
    if (rc != RHJ_OK) die(js.context(), "rhj_join_batch", rc);
    for (size_t i = 0; i < n; i++) {
        if (!pages[i]) continue;                                  // head stays nullptr -> isEmpty()
        results[i]->head = (bucket_info *)pages[i];               // next is already nullptr
        results[i]->capacity = counts[i];
        results[i]->size = counts[i];
    }
}

void Result::multiRadixHashJoin(JobScheduler &js, relation &relR, relation &relS)
{
    void *page = nullptr;
    uint64_t count = 0;
    int rc = rhj_join(js.context(), (const rhj_tuple *)relR.tuples, relR.num_tuples, (const rhj_tuple *)relS.tuples,
                      relS.num_tuples, nullptr, &page, &count);
    if (rc != RHJ_OK) die(js.context(), "rhj_join", rc);
    if (!page) return;
    if (head != nullptr) {                    // not fresh: append, keeping the existing pages valid
        addAll((bucket_info *)page, count);
        free(page);
        return;
    }
    head = (bucket_info *)page;               // next is already nullptr
    capacity = count;
    size = count;
}
