// rhj_query_dev.cpp -- device-resident execution of a Query (SURVEY §8f ranks 1-3 on the GPU).
// Same semantics as Query::execute's host path (rhj_query.cpp), but nothing except counts and the final
// SUMs crosses PCIe: the stored columns live in HBM (uploaded once), filters are rhj_col_filter, join inputs
// are built by rhj_gather_tuples, joins are rhj_join_dev, and the intermediate result is maintained by
// gathers.  An alias that is already part of the intermediate enters a join POSITION-CARRYING
// ({key = intermediate row, payload = value}): the pairs then name intermediate rows directly, which replaces
// both the de-duplication of structs.cpp:238-241 and update_intermediate's matching (intermediate.cpp:52-87).
#include <cstdio>
#include <cstdlib>
#include <map>
#include <mutex>
#include <vector>

#include "rhj_query.h"

namespace {

[[noreturn]] void die(rhj_ctx *ctx, const char *what, int rc)
{
    fprintf(stderr, "rhj: %s failed (%d): %s\n", what, rc, rhj_last_error(ctx));
    exit(EXIT_FAILURE);
}
#define OK(ctx, call) do { int rc_ = (call); if (rc_ != RHJ_OK) die(ctx, #call, rc_); } while (0)

struct DevArr {                                  // a uint64 (or 16-byte record) array in HBM
    rhj_ctx *ctx = nullptr;
    uint64_t *p = nullptr;
    uint64_t n = 0;
    DevArr() = default;
    DevArr(rhj_ctx *c, uint64_t bytes) : ctx(c)
    {
        void *q = nullptr;
        OK(ctx, rhj_dev_alloc(ctx, bytes ? bytes : 16, &q));
        p = (uint64_t *)q;
    }
    DevArr(const DevArr &) = delete;
    DevArr &operator=(const DevArr &) = delete;
    DevArr(DevArr &&o) noexcept { *this = std::move(o); }
    DevArr &operator=(DevArr &&o) noexcept
    {
        if (this != &o) { reset(); ctx = o.ctx; p = o.p; n = o.n; o.p = nullptr; o.n = 0; }
        return *this;
    }
    void reset() { if (p) rhj_dev_free(ctx, p); p = nullptr; n = 0; }
    ~DevArr() { reset(); }
    bool empty() const { return p == nullptr; }
};

// columns of the stored relations in HBM: uploaded once per process, shared by every context on the device
std::mutex g_cols_mu;
std::map<const uint64_t *, uint64_t *> g_cols;   // host column base -> device copy

const uint64_t *device_column(rhj_ctx *ctx, const relList &rel, uint64_t c)
{
    std::lock_guard<std::mutex> lk(g_cols_mu);
    auto it = g_cols.find(rel.values[c]);
    if (it != g_cols.end()) return it->second;
    void *d = nullptr;
    OK(ctx, rhj_dev_alloc(ctx, rel.num_tuples * 8, &d));
    OK(ctx, rhj_copy_h2d(ctx, d, rel.values[c], rel.num_tuples * 8));
    g_cols[rel.values[c]] = (uint64_t *)d;
    return (const uint64_t *)d;
}

// rows of an alias that survived its filters: device list, or "all rows" (p == nullptr, n = table size)
struct Rows {
    DevArr list;
    uint64_t n = 0;
    const uint64_t *ptr() const { return list.p; }
};

DevArr join_pairs(rhj_ctx *ctx, const DevArr &R, uint64_t nR, const DevArr &S, uint64_t nS, uint64_t &count)
{
    uint64_t cap = (nR > nS ? nR : nS) + 1024;
    for (;;) {
        DevArr out(ctx, cap * 16);
        int rc = rhj_join_dev(ctx, (const rhj_tuple *)R.p, nR, (const rhj_tuple *)S.p, nS, nullptr, (rhj_pair *)out.p, cap, &count);
        if (rc == RHJ_OK) { out.n = count; return out; }
        if (rc != RHJ_E_OVERFLOW) die(ctx, "rhj_join_dev", rc);
        cap = count;                                   // exact size is known now
    }
}

// every live intermediate column re-materialised through the positions `idx` (m of them)
void regather(rhj_ctx *ctx, std::vector<DevArr> &inter, const uint64_t *idx, uint64_t m)
{
    for (DevArr &col : inter) {
        if (col.empty()) continue;
        DevArr next(ctx, m * 8);
        OK(ctx, rhj_gather_u64(ctx, col.p, idx, m, next.p));
        next.n = m;
        col = std::move(next);
    }
}

}  // namespace

// Query::execute on the device.  Returns through this->filtered_out / proj[i].sum like the host path.
void Query::execute_device(JobScheduler &js, std::vector<relList> &relations)
{
    rhj_ctx *ctx = js.context();
    const size_t na = table.size();
    filtered_out = false;

    // ---- filters (Query.cpp:81-158) ------------------------------------------------------------
    std::vector<Rows> rows(na);
    for (size_t a = 0; a < na; a++) rows[a].n = relations[table[a]].num_tuples;
    for (const filter_info &f : filter) {
        const relList &rel = relations[table[f.table]];
        Rows &r = rows[f.table];
        DevArr out(ctx, r.n * 8);
        uint64_t m = 0;
        OK(ctx, rhj_col_filter(ctx, device_column(ctx, rel, f.column), r.ptr(), r.n, f.op, f.number, out.p, &m));
        if (m == 0) { filtered_out = true; return; }
        out.n = m;
        r.list = std::move(out);
        r.n = m;
    }

    // ---- join chain (Query.cpp:164-201) -------------------------------------------------------
    std::vector<DevArr> inter(na);                     // inter[a]: rowID of alias a per intermediate row
    uint64_t T = 0;                                    // intermediate rows
    for (const join_info &j : join) {
        const relList &rel1 = relations[table[j.table1]], &rel2 = relations[table[j.table2]];
        const uint64_t *c1 = device_column(ctx, rel1, j.column1), *c2 = device_column(ctx, rel2, j.column2);
        const bool in1 = !inter[j.table1].empty(), in2 = !inter[j.table2].empty();
        if (j.table1 == j.table2 || (in1 && in2)) {
            // a row filter: same-alias predicate (parse_table) or both aliases already joined (case 3)
            if (j.table1 == j.table2 && !in1) {
                Rows &r = rows[j.table1];
                DevArr pos(ctx, r.n * 8);
                uint64_t m = 0;
                OK(ctx, rhj_rows_filter_equal(ctx, c1, r.ptr(), c2, r.ptr(), r.n, pos.p, &m));
                if (m == 0) { filtered_out = true; return; }
                if (r.ptr() == nullptr) { pos.n = m; r.list = std::move(pos); }       // positions ARE the rowIDs
                else {
                    DevArr next(ctx, m * 8);
                    OK(ctx, rhj_gather_u64(ctx, r.ptr(), pos.p, m, next.p));
                    next.n = m;
                    r.list = std::move(next);
                }
                r.n = m;
                continue;
            }
            DevArr pos(ctx, T * 8);
            uint64_t m = 0;
            OK(ctx, rhj_rows_filter_equal(ctx, c1, inter[j.table1].p, c2, inter[j.table2].p, T, pos.p, &m));
            if (m == 0) { filtered_out = true; return; }
            regather(ctx, inter, pos.p, m);
            T = m;
            continue;
        }
        // an equi-join through the hot path.  Side already in the intermediate: position-carrying input.
        const uint64_t nR = in1 ? T : rows[j.table1].n, nS = in2 ? T : rows[j.table2].n;
        DevArr R(ctx, nR * 16), S(ctx, nS * 16);
        OK(ctx, rhj_gather_tuples(ctx, c1, in1 ? inter[j.table1].p : rows[j.table1].ptr(), nR, in1 ? 1 : 0, (rhj_tuple *)R.p));
        OK(ctx, rhj_gather_tuples(ctx, c2, in2 ? inter[j.table2].p : rows[j.table2].ptr(), nS, in2 ? 1 : 0, (rhj_tuple *)S.p));
        uint64_t m = 0;
        DevArr pairs = join_pairs(ctx, R, nR, S, nS, m);                              // <-- rhj_join_dev
        if (m == 0) { filtered_out = true; return; }
        DevArr kr(ctx, m * 8), ks(ctx, m * 8);
        OK(ctx, rhj_pairs_split(ctx, (const rhj_pair *)pairs.p, m, kr.p, ks.p));
        kr.n = ks.n = m;
        if (!in1 && !in2) {                            // case 1: the pairs become the two columns
            // (a join between two aliases that are both new while OTHER aliases are already joined would need a
            //  cross product; the reference's update_intermediate drops the older columns in that case
            //  (intermediate.cpp:147-162: only table1/table2 of intermediate_upd are filled) -- same here)
            for (DevArr &c : inter) c.reset();
            inter[j.table1] = std::move(kr);
            inter[j.table2] = std::move(ks);
        } else if (in1) {                              // case 2: keyR = intermediate row, keyS = new alias' rowID
            regather(ctx, inter, kr.p, m);
            inter[j.table2] = std::move(ks);
        } else {
            regather(ctx, inter, ks.p, m);
            inter[j.table1] = std::move(kr);
        }
        T = m;
    }

    // ---- SUM projections (Query.cpp:66-74,198-200) -------------------------------------------
    for (proj_info &p : proj) {
        const uint64_t *col = device_column(ctx, relations[table[p.table]], p.column);
        // an alias that is not part of the intermediate (never joined, or dropped when a later join linked two new
        // aliases) sums to 0, as in the reference (Query.cpp:198-200 over an empty intermediate[p.table])
        uint64_t sum = 0;
        if (!inter[p.table].empty()) OK(ctx, rhj_sum_gather(ctx, col, inter[p.table].p, T, &sum));
        p.sum = sum;
    }
}
