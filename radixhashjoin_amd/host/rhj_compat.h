/*
 * rhj_compat.h -- C++ host mirror of the reference's hot-path surface, implemented over the
 * C-ABI of include/rhj.h (librhj_hip.so).  A caller written against the reference's
 *     structs.h (tuple / relation / relation_info), Result.h (key_tuple / bucket_info / Result),
 *     JobScheduler.h (Job / HistogramJob / PartitionJob / JoinJob / JobScheduler)
 * compiles against this header unchanged: same type names, member names, argument order and
 * meaning (tests/test_integration_build.py compiles the reference's own Query.cpp, intermediate.cpp,
 * MainScheduler.cpp and join.cpp against it).  `Job` carries one extra public member, `gpu`, which the
 * reference's callers never touch.  What differs is WHERE the work runs:
 *
 *   reference (CPU, pthreads)                              here (MI355X)
 *   ----------------------------------------------------   --------------------------------------------
 *   JobScheduler::init(n)  spawns n worker threads          opens one rhj_ctx (HIP stream + HBM workspace)
 *   JobScheduler::schedule queues a Job for a worker        runs the job on the caller; its kernels are
 *                                                           asynchronous on the context's stream
 *   JobScheduler::barrier  waits for the queue to drain     rhj_sync (stream synchronize)
 *   Result::multiRadixHashJoin  2x hash_relation + 256       ONE rhj_join call: histogram, prefix, scatter
 *                          JoinJobs + serial page concat    (1-2 passes), LDS bucket join, one result page
 *
 * Error behaviour: the reference has no error channel (void returns, assert / exit(EXIT_FAILURE));
 * the mirror keeps that: a failing C-ABI call prints rhj_last_error() and exits.  There is no CPU
 * fallback.
 */
#ifndef RHJ_COMPAT_H
#define RHJ_COMPAT_H

#include <cstddef>
#include <cstdint>
#include <unordered_map>
#include <unordered_set>
#include <vector>

#include "../../include/rhj.h"

#ifndef NUM_OF_THREADS
#define NUM_OF_THREADS 8        /* reference JobScheduler.h:11; kept as the nominal worker count */
#endif

struct relation;
struct relation_info;
struct relList;
struct Result;
struct WorkerPool;

/* ---------------------------------------------------------------- JobScheduler.h surface ---------- */

class Job {                                   /* JobScheduler.h:16-28 */
public:
    Job() = default;
    virtual ~Job() = default;
    virtual void init(void *arg);             /* called with the scheduler's argument before run() */
    virtual int run() = 0;
    rhj_ctx *gpu = nullptr;                   /* set by JobScheduler::schedule: the context the job's kernels run on */
};

class JobScheduler {                          /* JobScheduler.h:87-124 */
    rhj_ctx *ctx_ = nullptr;
    size_t num_of_threads = 0;
    void *job_arg_ = nullptr;                 /* what Job::init receives (the reference passes threadWork's argument) */
    WorkerPool *pool_ = nullptr;              /* queue + host threads; only after init(n, start_routine) */
public:
    JobScheduler() = default;
    virtual ~JobScheduler() = default;

    /* JobScheduler.cpp:22-64: the worker loop.  Takes jobs off this scheduler's queue until stop(); every job gets
       job->init(arg) before run() and is deleted afterwards.  Only meaningful after init(n, start_routine). */
    void threadWork(void *arg);
    /* JobScheduler.cpp:67-82: n host threads, each running start_routine(this) (which is expected to call
       threadWork, as mainThreadWork does, MainScheduler.cpp:6-14).  Opens no GPU context of its own. */
    bool init(size_t num_of_threads, void *start_routine(void *));
    virtual bool init(size_t num_of_threads); /* JobScheduler.cpp:84-86: here: rhj_init on $RHJ_DEVICE (default 0) */
    bool destroy();                           /* JobScheduler.cpp:89-97  */
    void barrier();                           /* JobScheduler.cpp:103-122: all scheduled work has completed */
    int schedule(Job *job);                   /* JobScheduler.cpp:125-137: takes ownership, deletes the job */
    void stop();                              /* JobScheduler.cpp:140-146 */

    rhj_ctx *context() const { return ctx_; } /* the GPU context this scheduler dispatches to */
    void set_job_arg(void *arg) { job_arg_ = arg; }
};

class HistogramJob : public Job {             /* JobScheduler.h:30-43, body JobScheduler.cpp:149-155 */
    size_t *histogram;
    relation &rel;
    size_t twoInLSB, start, end;
    int run() override;
public:
    HistogramJob(size_t *histogram, relation &rel, size_t twoInLSB, size_t start, size_t end);
};

class PartitionJob : public Job {             /* JobScheduler.h:45-59, body JobScheduler.cpp:162-177 */
    size_t *tuples;
    relation &rel;
    size_t twoInLSB, start, end;
    size_t *sumHistogram, *histogram;
    int run() override;
public:
    PartitionJob(size_t *tuples, relation &rel, size_t twoInLSB, size_t start, size_t end, size_t *sumHistogram,
                 size_t *histogram);
};

class JoinJob : public Job {                  /* JobScheduler.h:61-75, body JobScheduler.cpp:186-192 */
    Result &result;
    relation_info *relShashed, *relRhashed;
    size_t begS, begR, histS, histR;
    int run() override;
public:
    JoinJob(Result &result, relation_info *relShashed, relation_info *relRhashed, size_t begS, size_t begR,
            size_t histS, size_t histR);
};

/* ---------------------------------------------------------------- structs.h surface (hot path) ---- */

struct tuple {                                /* structs.h:33-36: key = rowID, payload = join value */
    uint64_t key;
    uint64_t payload;
};

struct relation {                             /* structs.h:38-49 */
    tuple *tuples;
    uint64_t num_tuples;
    /* query-layer side of the boundary (rhj_query.cpp): build the AoS input of a join from the filtered
       rowIDs of an alias or, once the alias is part of the intermediate, from its DISTINCT rowIDs
       (structs.cpp:217-243) */
    void create_relation(uint64_t join_table, relList &rel, uint64_t column_number,
                         std::unordered_map<uint64_t, std::unordered_set<uint64_t> > &filtered,
                         std::vector<uint64_t> &inter);
    void foo(const relList &rel, size_t column_number, const std::unordered_set<uint64_t> &uniqueValues);
    ~relation();                              /* delete[] tuples, structs.cpp:210-212 */
};

struct relation_info {                        /* structs.h:51-58 */
    relation tuples;                          /* R': bucket-contiguous copy */
    size_t *histogram;                        /* twoInLSB counts */
    void hash_relation(JobScheduler &js, relation &rel, size_t twoInLSB);   /* structs.cpp:144-204 */
    ~relation_info();                         /* delete[] histogram, structs.cpp:206-208 */
};

/* ---------------------------------------------------------------- Result.h surface ---------------- */

struct key_tuple {                            /* Result.h:9-12 */
    uint64_t keyR;
    uint64_t keyS;
};

struct bucket_info {                          /* Result.h:14-17: page header, pairs follow in the same block */
    bucket_info *next;
};

struct Result {                               /* Result.h:19-38 */
    size_t capacity;                          /* pairs per page */
    size_t size;                              /* pairs in the head page; every other page is full */
    bucket_info *head;                        /* newest page; nullptr = empty result */

    Result();
    ~Result();
    bool isEmpty();
    void add_result(uint64_t keyR, uint64_t keyS);
    void addAll(bucket_info *node, size_t size);
    void multiRadixHashJoin(JobScheduler &js, relation &relR, relation &relS);
    /* NOT in the reference: n joins in one call (rhj_join_batch, sixteen small joins per GPU launch).  results[i] must be
       fresh Results; afterwards each is what results[i]->multiRadixHashJoin(js, *R[i], *S[i]) would have left.  Used by the
       level-by-level query executor of rhj_query.cpp (RHJ_QUERY_MODE=batch). */
    static void multiRadixHashJoinBatch(JobScheduler &js, size_t n, relation *const *R, relation *const *S, Result *const *results);
    void join_buckets(relation_info *small, relation_info *big, size_t begSmall, size_t begBig, size_t histSmall,
                      size_t histBig, bool orderFlag);
private:
    friend class JoinJob;
    void join_slices(rhj_ctx *ctx, const tuple *small, size_t nSmall, const tuple *big, size_t nBig, bool orderFlag);
};

static_assert(sizeof(tuple) == sizeof(rhj_tuple) && sizeof(key_tuple) == sizeof(rhj_pair), "C-ABI layouts");

#endif /* RHJ_COMPAT_H */
