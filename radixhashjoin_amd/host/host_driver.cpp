// host_driver.cpp -- exercises the reference-surface mirror exactly the way the reference's own
// code does, for tests/test_gpu_host_mirror.py.
//   host_driver direct R.bin S.bin out.bin   : Result::multiRadixHashJoin (the drop-in, Query.cpp:185-186)
//   host_driver staged R.bin S.bin out.bin   : replays Result.cpp:90-124 through the job classes:
//                                              hash_relation x2, one JoinJob per bucket, barrier, addAll
//   host_driver jobs   R.bin out.bin         : HistogramJob + PartitionJob over 8 row ranges (structs.cpp:111-134)
// Relations are raw arrays of 16-byte {key,payload}; outputs are raw arrays.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "rhj_compat.h"

static void load(const char *path, relation &r)
{
    FILE *f = fopen(path, "rb");
    if (!f) { perror(path); exit(2); }
    fseek(f, 0, SEEK_END);
    long bytes = ftell(f);
    fseek(f, 0, SEEK_SET);
    r.num_tuples = (uint64_t)bytes / sizeof(tuple);
    r.tuples = new tuple[r.num_tuples ? r.num_tuples : 1];
    if (r.num_tuples && fread(r.tuples, sizeof(tuple), r.num_tuples, f) != r.num_tuples) { perror("fread"); exit(2); }
    fclose(f);
}

// walk the pages the way every consumer of the reference does (intermediate.cpp:151-179)
static void dump(const char *path, Result &res)
{
    FILE *f = fopen(path, "wb");
    size_t sz = res.size;
    for (bucket_info *n = res.head; n; n = n->next) {
        fwrite(&n[1], sizeof(key_tuple), sz, f);
        sz = res.capacity;
    }
    fclose(f);
}

int main(int argc, char **argv)
{
    if (argc < 4) { fprintf(stderr, "usage: see header comment\n"); return 2; }
    const char *mode = argv[1];
    JobScheduler js;
    js.init(NUM_OF_THREADS);
    if (!strcmp(mode, "jobs")) {
        relation R;
        load(argv[2], R);
        const size_t nb = 256, T = NUM_OF_THREADS;
        std::vector<size_t> start(T), end(T);
        const size_t q = R.num_tuples / T;
        size_t r = R.num_tuples % T;
        start[0] = 0; end[0] = q;
        for (size_t i = 1; i < T; i++) { start[i] = end[i - 1]; end[i] = start[i] + q; if (r) { end[i]++; r--; } }
        std::vector<std::vector<size_t>> hist(T, std::vector<size_t>(nb, 0)), sum(T, std::vector<size_t>(nb, 0)), idx(T);
        for (size_t i = 0; i < T; i++) js.schedule(new HistogramJob(hist[i].data(), R, nb, start[i], end[i]));
        js.barrier();
        for (size_t i = 0; i < T; i++) {
            idx[i].resize(end[i] - start[i] + 1);
            js.schedule(new PartitionJob(idx[i].data(), R, nb, start[i], end[i], sum[i].data(), hist[i].data()));
        }
        js.barrier();
        FILE *f = fopen(argv[3], "wb");
        for (size_t i = 0; i < T; i++) {
            uint64_t n = end[i] - start[i];
            fwrite(&n, 8, 1, f);
            fwrite(hist[i].data(), sizeof(size_t), nb, f);
            fwrite(sum[i].data(), sizeof(size_t), nb, f);
            fwrite(idx[i].data(), sizeof(size_t), n, f);
        }
        fclose(f);
    } else {
        if (argc < 5) return 2;
        relation R, S;
        load(argv[2], R);
        load(argv[3], S);
        Result res;
        if (!strcmp(mode, "direct")) {
            res.multiRadixHashJoin(js, R, S);
        } else {
            const size_t nb = 256;
            relation_info hr, hs;
            hr.hash_relation(js, R, nb);
            hs.hash_relation(js, S, nb);
            Result *part = new Result[nb];
            size_t begR = 0, begS = 0;
            for (size_t b = 0; b < nb; b++) {
                if (hr.histogram[b] && hs.histogram[b])
                    js.schedule(new JoinJob(part[b], &hs, &hr, begS, begR, hs.histogram[b], hr.histogram[b]));
                begR += hr.histogram[b];
                begS += hs.histogram[b];
            }
            js.barrier();
            for (size_t b = 0; b < nb; b++) {
                if (part[b].isEmpty()) continue;
                bucket_info *n = part[b].head;
                res.addAll(n, part[b].size);
                for (n = n->next; n; n = n->next) res.addAll(n, part[b].capacity);
            }
            delete[] part;
        }
        size_t matches = 0, sz = res.size;
        for (bucket_info *n = res.head; n; n = n->next) { matches += sz; sz = res.capacity; }
        printf("%s matches=%zu head=%s capacity=%zu size=%zu\n", mode, matches, res.isEmpty() ? "null" : "set",
               res.capacity, res.size);
        dump(argv[4], res);
    }
    js.stop();
    js.destroy();
    return 0;
}
