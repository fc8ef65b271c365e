// harness.cpp -- drives the host threads of rhj_join (rhj_api.hip, compiled here with g++ -fsanitize=thread over the fake
// runtime of tsan/hip/hip_runtime.h) the way the reference's query threads would: several threads, a context each
// (MainScheduler.cpp:6-14), every host path of the drop-in call:
//   * small joins (H2D, H2D, one launch, one synchronise; the result lands in pinned memory),
//   * the plain large path (six stager workers filling a ring of pinned buffers that the copy stream drains, four page
//     pre-faulters, one D2H),
//   * the pipelined path (the same stager per S chunk, a downloader thread bringing finished ranges of pairs home on a third
//     stream while later chunks are still being uploaded).
// The fake "bucket join" appends 2^20 pairs {i, ~i} per launch, so every page that comes back can be checked byte for byte:
// a range the downloader missed, copied early or copied twice shows up here, a data race shows up in ThreadSanitizer's report.
// Exit code 0 = every page as expected; TSan makes the exit code 66 when it has reported anything (TSAN_OPTIONS exitcode).
#include "../../../include/rhj.h"

#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

static const uint64_t FAKE_PAIRS = 1ull << 20;

static bool page_ok(const void *page, uint64_t count, const char *what)
{
    if (!page) { fprintf(stderr, "%s: no page\n", what); return false; }
    const uint64_t *w = (const uint64_t *)page;
    if (w[0] != 0) { fprintf(stderr, "%s: next pointer not null\n", what); return false; }
    for (uint64_t i = 0; i < count; i++)
        if (w[1 + 2 * i] != i || w[2 + 2 * i] != ~i) { fprintf(stderr, "%s: pair %llu wrong\n", what, (unsigned long long)i); return false; }
    return true;
}

static std::vector<rhj_tuple> relation(uint64_t n)
{
    std::vector<rhj_tuple> t(n);
    for (uint64_t i = 0; i < n; i++) { t[i].key = i; t[i].payload = i * 0x9E3779B97F4A7C15ull; }
    return t;
}

int main()
{
    setenv("RHJ_PIPE_MIN_CHUNK", "4194304", 1);           // pipelining from 16 Mi tuples of S on: four 64 MiB chunks, each staged
    std::atomic<bool> ok{true};
    auto query_thread = [&](int id, uint64_t nR, uint64_t nS, int reps, uint64_t joins_per_call) {
        rhj_ctx *ctx = nullptr;
        if (rhj_init(0, &ctx) != RHJ_OK) { ok = false; return; }
        const std::vector<rhj_tuple> R = relation(nR), S = relation(nS);
        for (int r = 0; r < reps; r++) {
            void *page = nullptr;
            uint64_t count = 0;
            const int rc = rhj_join(ctx, R.data(), nR, S.data(), nS, nullptr, &page, &count);
            char what[96];
            snprintf(what, sizeof what, "thread %d call %d (%llu x %llu)", id, r, (unsigned long long)nR, (unsigned long long)nS);
            if (rc != RHJ_OK) { fprintf(stderr, "%s: rc %d: %s\n", what, rc, rhj_last_error(ctx)); ok = false; }
            else if (count != joins_per_call * FAKE_PAIRS) { fprintf(stderr, "%s: count %llu\n", what, (unsigned long long)count); ok = false; }
            else if (!page_ok(page, count, what)) ok = false;
            free(page);
        }
        rhj_destroy(ctx);
    };
    std::vector<std::thread> th;
    // two pipelined callers (K = 4 chunks of S: 4 joins per call), two plain large callers, four small-join callers
    th.emplace_back(query_thread, 0, 6u << 20, 16u << 20, 2, 4);
    th.emplace_back(query_thread, 1, 5u << 20, 17u << 20, 1, 4);
    th.emplace_back(query_thread, 2, 5u << 20, 5u << 20, 2, 1);
    th.emplace_back(query_thread, 3, 9u << 20, 4u << 20, 1, 1);
    for (int k = 0; k < 4; k++) th.emplace_back(query_thread, 4 + k, 1500 + 700 * k, 9000 + 3000 * k, 16, 1);
    // two callers of rhj_join_batch: 40 small joins per call (three launches of <= 16), inputs staged and pages filled by the
    // context's helper threads
    auto batch_thread = [&](int id) {
        rhj_ctx *ctx = nullptr;
        if (rhj_init(0, &ctx) != RHJ_OK) { ok = false; return; }
        std::vector<std::vector<rhj_tuple>> rel;
        std::vector<rhj_join_desc> joins;
        for (int j = 0; j < 40; j++) { rel.push_back(relation(900 + 531 * j)); rel.push_back(relation(20000 + 997 * j)); }
        for (int j = 0; j < 40; j++) joins.push_back(rhj_join_desc{rel[2 * j].data(), rel[2 * j].size(), rel[2 * j + 1].data(), rel[2 * j + 1].size()});
        for (int r = 0; r < 3; r++) {
            std::vector<void *> pages(joins.size());
            std::vector<uint64_t> counts(joins.size());
            const int rc = rhj_join_batch(ctx, (uint32_t)joins.size(), joins.data(), pages.data(), counts.data());
            if (rc != RHJ_OK) { fprintf(stderr, "batch thread %d: rc %d: %s\n", id, rc, rhj_last_error(ctx)); ok = false; }
            for (size_t j = 0; j < joins.size(); j++) {
                char what[64];
                snprintf(what, sizeof what, "batch thread %d call %d join %zu", id, r, j);
                if (rc == RHJ_OK && (counts[j] != FAKE_PAIRS || !page_ok(pages[j], counts[j], what))) ok = false;
                free(pages[j]);
            }
        }
        rhj_destroy(ctx);
    };
    th.emplace_back(batch_thread, 20);
    th.emplace_back(batch_thread, 21);
    for (std::thread &t : th) t.join();
    puts(ok ? "tsan harness: every page as expected" : "tsan harness: FAILED");
    return ok ? 0 : 1;
}
