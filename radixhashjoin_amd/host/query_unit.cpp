// query_unit.cpp -- CPU-only unit checks of the query layer's host logic (no GPU call is made):
// query text parser, the three update_intermediate cases against a brute-force restatement of
// the reference's semantics (intermediate.cpp:52-87,146-183), Result page bookkeeping.
// Exit code 0 = all checks passed.  Run by tests/test_host_logic.py.
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <string>
#include <vector>

#include "rhj_query.h"

static int failures = 0;
#define CHECK(c) do { if (!(c)) { fprintf(stderr, "CHECK failed %s:%d: %s\n", __FILE__, __LINE__, #c); failures++; } } while (0)

typedef std::vector<std::vector<uint64_t> > Inter;

// reference semantics, written the slow obvious way
static Inter brute(const Inter &in, const std::vector<key_tuple> &pairs, uint64_t t1, uint64_t t2)
{
    Inter out(in.size());
    const bool e1 = in[t1].empty(), e2 = in[t2].empty();
    if (e1 && e2) {
        for (const key_tuple &p : pairs) { out[t1].push_back(p.keyR); out[t2].push_back(p.keyS); }
    } else if (e1 || e2) {
        const uint64_t have = e1 ? t2 : t1, fresh = e1 ? t1 : t2;
        for (const key_tuple &p : pairs)
            for (size_t e = 0; e < in[have].size(); e++)
                if (in[have][e] == (e1 ? p.keyS : p.keyR)) {
                    for (size_t i = 0; i < in.size(); i++) if (!in[i].empty()) out[i].push_back(in[i][e]);
                    out[fresh].push_back(e1 ? p.keyR : p.keyS);
                }
    } else {
        for (const key_tuple &p : pairs)
            for (size_t e = 0; e < in[t1].size(); e++)
                if (in[t1][e] == p.keyR && in[t2][e] == p.keyS)
                    for (size_t i = 0; i < in.size(); i++) if (!in[i].empty()) out[i].push_back(in[i][e]);
    }
    return out;
}

static std::vector<std::vector<uint64_t> > rows_sorted(const Inter &x)
{
    size_t n = 0;
    for (const auto &c : x) n = std::max(n, c.size());
    std::vector<std::vector<uint64_t> > rows(n);
    for (size_t e = 0; e < n; e++) for (const auto &c : x) rows[e].push_back(c.empty() ? ~0ull : c[e]);
    std::sort(rows.begin(), rows.end());
    return rows;
}

int main()
{
    {   // parser (Query.cpp:10-63): tables | predicates | projections
        Query q(std::string("3 0 1|0.2=1.0&0.1=2.0&0.2>3499|1.2 0.1"));
        CHECK(q.table.size() == 3 && q.table[0] == 3 && q.table[1] == 0 && q.table[2] == 1);
        CHECK(q.join.size() == 2 && q.join[0].table1 == 0 && q.join[0].column1 == 2 && q.join[0].table2 == 1 && q.join[0].column2 == 0);
        CHECK(q.join[1].table1 == 0 && q.join[1].column1 == 1 && q.join[1].table2 == 2 && q.join[1].column2 == 0);
        CHECK(q.filter.size() == 1 && q.filter[0].table == 0 && q.filter[0].column == 2 && q.filter[0].op == '>' && q.filter[0].number == 3499);
        CHECK(q.proj.size() == 2 && q.proj[0].table == 1 && q.proj[0].column == 2 && q.proj[1].table == 0 && q.proj[1].column == 1);
        Query q2(std::string("5 0|0.2=1.0&0.3=9881|1.1 0.2 1.0"));
        CHECK(q2.join.size() == 1 && q2.filter.size() == 1 && q2.filter[0].op == '=' && q2.filter[0].number == 9881 && q2.proj.size() == 3);
        Query q3(std::string("6 1 12|0.1=1.0&1.0=2.2&0.0<62236|1.0"));
        CHECK(q3.filter[0].op == '<' && q3.join[1].table1 == 1 && q3.join[1].table2 == 2 && q3.join[1].column2 == 2);
        q3.filtered_out = true;
        CHECK(q3.result_line() == "NULL");
    }
    {   // Result pages (Result.cpp:10-35,78-84): 8191 pairs per 128 KiB page, LIFO, only head partial
        Result r;
        CHECK(r.isEmpty() && r.capacity == 8191 && r.size == 8191);
        for (uint64_t i = 0; i < 20000; i++) r.add_result(i, i * 3);
        size_t pages = 0, n = 0, sz = r.size;
        for (bucket_info *p = r.head; p; p = p->next) { pages++; n += sz; sz = r.capacity; }
        CHECK(pages == 3 && n == 20000 && r.size == 20000 - 2 * 8191);
        Result all;
        bucket_info *node = r.head;
        all.addAll(node, r.size);
        for (node = node->next; node; node = node->next) all.addAll(node, r.capacity);
        size_t m = 0; sz = all.size;
        for (bucket_info *p = all.head; p; p = p->next) { m += sz; sz = all.capacity; }
        CHECK(m == 20000);
    }
    std::mt19937_64 rng(7);
    for (int trial = 0; trial < 200; trial++) {   // update_intermediate vs brute force, all three cases
        const size_t na = 4, rows = rng() % 60;
        const int mode = trial % 3;               // 0: neither alias joined, 1: one, 2: both
        Inter in(na);
        uint64_t t1 = 0, t2 = 1;
        if (mode >= 1) for (size_t e = 0; e < rows + 1; e++) { in[0].push_back(rng() % 8); in[2].push_back(rng() % 5); }
        if (mode == 2) for (size_t e = 0; e < rows + 1; e++) in[1].push_back(rng() % 8);
        if (mode == 1 && (trial & 4)) { t1 = 1; t2 = 0; }      // the joined alias may be on either side
        // result pairs: unique rowIDs per side combinations (a join of de-duplicated inputs never repeats a pair)
        std::vector<key_tuple> pairs;
        for (uint64_t r = 0; r < 8; r++) for (uint64_t s = 0; s < 8; s++) if (rng() % 3 == 0) pairs.push_back(key_tuple{r, s});
        if (pairs.empty()) pairs.push_back(key_tuple{1, 1});
        Result res;
        for (const key_tuple &p : pairs) res.add_result(p.keyR, p.keyS);
        join_info j(t1, 0, t2, 0);
        Inter got = in;
        update_intermediate(got, res, j);
        Inter exp = brute(in, pairs, t1, t2);
        CHECK(rows_sorted(got) == rows_sorted(exp));
    }
    if (failures) { fprintf(stderr, "%d checks failed\n", failures); return 1; }
    printf("query_unit: all checks passed\n");
    return 0;
}
