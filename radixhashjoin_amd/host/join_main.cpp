// join_main.cpp -- the reference's CLI protocol (join.cpp:11-62) on the GPU engine:
//   stdin = relation file paths, "Done", then batches of queries separated by lines "F";
//   stdout = one line of SUMs (or NULLs) per query, in input order.
//   cd tests/golden && cat small/small.init small/small.work | ../../radixhashjoin_amd/host/join_gpu
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <iostream>
#include <string>
#include <vector>

#include "rhj_query.h"

int main()
{
    std::vector<relList> relations;
    std::string line;
    while (std::getline(std::cin, line) && line != "Done") {
        if (line.empty()) continue;
        std::vector<char> path(line.begin(), line.end());
        path.push_back('\0');
        relations.emplace_back(path.data());
    }
    std::vector<std::vector<Query> > batches(1);
    while (std::getline(std::cin, line)) {
        if (line.empty()) continue;
        if (line == "F") { if (!batches.back().empty()) batches.emplace_back(); continue; }
        batches.back().emplace_back(line);
    }
    const char *mode = getenv("RHJ_QUERY_MODE");
    if (mode && std::string(mode) == "batch") {
        // one thread, one GPU context; every batch of queries level by level, the joins of a level sixteen per GPU launch
        JobScheduler js;
        js.init(NUM_OF_THREADS);
        for (auto &batch : batches) Query::execute_batch(js, batch, relations);
        js.stop();
        js.destroy();
        for (auto &batch : batches)
            for (const Query &q : batch) q.print();
        for (relList &r : relations) r.destroy();
        return 0;
    }
    MainScheduler ms;
    ms.init(NUM_OF_THREADS);                 // 8 query threads, each with a private JobScheduler = private GPU context
    for (auto &batch : batches)
        for (Query &q : batch) ms.schedule(new QueryJob(q, relations));
    ms.stop();
    ms.destroy();
    for (auto &batch : batches)
        for (const Query &q : batch) q.print();
    for (relList &r : relations) r.destroy();
    return 0;
}
