// rhj_query.cpp -- query layer over the GPU join (see rhj_query.h).  New code; semantics follow
// the reference (Query.cpp, intermediate.cpp, structs.cpp:17-84,217-243, MainScheduler.cpp).
#include "rhj_query.h"

#include <algorithm>
#include <cassert>
#include <cctype>
#include <cstdio>
#include <cstdlib>
#include <fcntl.h>
#include <iostream>
#include <mutex>
#include <sstream>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

using std::unordered_map;
using std::unordered_set;
using std::vector;

typedef unordered_map<uint64_t, unordered_set<uint64_t> > FilteredRows;

// ------------------------------------------------------------------------------------------------
// relList: column file -> mmap'd columns + min/max/distinct (structs.cpp:17-72)
// ------------------------------------------------------------------------------------------------
relList::relList(char *filename)
{
    const int fd = open(filename, O_RDONLY);
    if (fd < 0) { perror(filename); exit(EXIT_FAILURE); }
    struct stat st;
    fstat(fd, &st);
    uint64_t *base = (uint64_t *)mmap(nullptr, (size_t)st.st_size, PROT_READ, MAP_PRIVATE, fd, 0);
    if (base == MAP_FAILED) { perror("mmap"); exit(EXIT_FAILURE); }
    close(fd);
    num_tuples = base[0];
    num_columns = base[1];
    if ((uint64_t)st.st_size != (num_tuples * num_columns + 2) * sizeof(uint64_t)) {
        fprintf(stderr, "%s: size does not match its header\n", filename);
        exit(EXIT_FAILURE);
    }
    values = new uint64_t *[num_columns];
    col_min = new uint64_t[num_columns];
    col_max = new uint64_t[num_columns];
    distinct = new uint64_t[num_columns]();
    for (uint64_t c = 0; c < num_columns; c++) {
        const uint64_t *col = base + 2 + c * num_tuples;
        values[c] = const_cast<uint64_t *>(col);
        uint64_t lo = num_tuples ? col[0] : 0, hi = lo;
        for (uint64_t r = 1; r < num_tuples; r++) { lo = std::min(lo, col[r]); hi = std::max(hi, col[r]); }
        col_min[c] = lo;
        col_max[c] = hi;
        // distinct count: presence bitmap over [min,max] like the reference when the range is small,
        // a sorted copy otherwise (the reference's vector<bool>(max-min+1) cannot hold wide value ranges)
        if (num_tuples && hi - lo < (uint64_t)1 << 27) {
            vector<bool> seen(hi - lo + 1, false);
            uint64_t d = 0;
            for (uint64_t r = 0; r < num_tuples; r++)
                if (!seen[col[r] - lo]) { seen[col[r] - lo] = true; d++; }
            distinct[c] = d;
        } else if (num_tuples) {
            vector<uint64_t> tmp(col, col + num_tuples);
            std::sort(tmp.begin(), tmp.end());
            distinct[c] = (uint64_t)(std::unique(tmp.begin(), tmp.end()) - tmp.begin());
        }
    }
}

void relList::destroy()
{
    munmap(values[0] - 2, (num_tuples * num_columns + 2) * sizeof(uint64_t));
    delete[] values; delete[] col_min; delete[] col_max; delete[] distinct;
}

void relList_stats::fill(relList &r)
{
    size = r.num_tuples;
    low = new uint64_t[r.num_columns];
    max = new uint64_t[r.num_columns];
    distinct = new uint64_t[r.num_columns];
    for (uint64_t c = 0; c < r.num_columns; c++) { low[c] = r.col_min[c]; max[c] = r.col_max[c]; distinct[c] = r.distinct[c]; }
}

// ------------------------------------------------------------------------------------------------
// relation builders: the input side of the hot-path boundary (structs.cpp:217-243)
// ------------------------------------------------------------------------------------------------
void relation::foo(const relList &rel, size_t column_number, const unordered_set<uint64_t> &rows)
{
    num_tuples = rows.size();
    tuples = new tuple[num_tuples ? num_tuples : 1];
    size_t i = 0;
    for (uint64_t rowid : rows) { tuples[i].key = rowid; tuples[i].payload = rel.values[column_number][rowid]; i++; }
}

void relation::create_relation(uint64_t join_table, relList &rel, uint64_t column_number, FilteredRows &filtered,
                               vector<uint64_t> &inter)
{
    if (inter.empty()) { foo(rel, column_number, filtered.find(join_table)->second); return; }
    unordered_set<uint64_t> uniq(inter.begin(), inter.end());        // an alias already joined contributes each rowID once
    foo(rel, column_number, uniq);
}

// ------------------------------------------------------------------------------------------------
// Query text: "t0 t1 ...|a.c=b.d&a.c>N&...|a.c b.d ..."  (Query.cpp:10-63)
// ------------------------------------------------------------------------------------------------
join_info::join_info(uint64_t t1, uint64_t c1, uint64_t t2, uint64_t c2) : table1(t1), column1(c1), table2(t2), column2(c2) {}
filter_info::filter_info(uint64_t t, uint64_t c, int o, uint64_t n) : table(t), column(c), op(o), number(n) {}
proj_info::proj_info(uint64_t t, uint64_t c) : table(t), column(c), sum(0) {}

static uint64_t take_number(const std::string &s, size_t &p)
{
    uint64_t v = 0;
    while (p < s.size() && isdigit((unsigned char)s[p])) v = v * 10 + (uint64_t)(s[p++] - '0');
    return v;
}

Query::Query(const std::string &line) : filtered_out(false), text_(line) { parse_all(); }

Query::Query(int ch) : filtered_out(false)
{
    // the reference parses with getchar() as it goes; here the rest of the line is read first
    text_.push_back((char)ch);
    for (int c = getchar(); c != EOF && c != '\n'; c = getchar()) text_.push_back((char)c);
    parse_all();
}

void Query::parse_all()
{
    pos_ = 0;
    read_relations(0);
    read_predicates();
    read_projections();
    stats.resize(table.size());
}

bool Query::read_relations(int)
{
    while (pos_ < text_.size() && text_[pos_] != '|') {
        if (isdigit((unsigned char)text_[pos_])) table.push_back(take_number(text_, pos_));
        else pos_++;
    }
    pos_++;                                                             // '|'
    return false;
}

void Query::read_predicates()
{
    while (pos_ < text_.size() && text_[pos_] != '|') {
        const uint64_t t1 = take_number(text_, pos_); pos_++;           // '.'
        const uint64_t c1 = take_number(text_, pos_);
        const int op = text_[pos_++];
        const uint64_t x = take_number(text_, pos_);
        if (pos_ < text_.size() && text_[pos_] == '.') {                // a.c = b.d : join
            pos_++;
            join.emplace_back(t1, c1, x, take_number(text_, pos_));
        } else {
            filter.emplace_back(t1, c1, op, x);
        }
        if (pos_ < text_.size() && text_[pos_] == '&') pos_++;
    }
    pos_++;
}

void Query::read_projections()
{
    while (pos_ < text_.size()) {
        if (!isdigit((unsigned char)text_[pos_])) { pos_++; continue; }
        const uint64_t t = take_number(text_, pos_); pos_++;
        proj.emplace_back(t, take_number(text_, pos_));
    }
}

// ------------------------------------------------------------------------------------------------
// filters (Query.cpp:81-158): surviving rowIDs per alias; true = some alias has no row left
// ------------------------------------------------------------------------------------------------
bool Query::run_filters(vector<relList> &relations, FilteredRows &filtered)
{
    for (uint64_t a = 0; a < table.size(); a++) stats[a].fill(relations[table[a]]);
    vector<vector<uint64_t> > rows(table.size());
    vector<bool> touched(table.size(), false);
    for (const filter_info &f : filter) {
        const relList &rel = relations[table[f.table]];
        const uint64_t *col = rel.values[f.column];
        vector<uint64_t> &cur = rows[f.table];
        auto keep = [&](uint64_t v) { return f.op == '>' ? v > f.number : f.op == '<' ? v < f.number : v == f.number; };
        if (!touched[f.table]) {
            for (uint64_t r = 0; r < rel.num_tuples; r++) if (keep(col[r])) cur.push_back(r);
            touched[f.table] = true;
        } else {
            size_t w = 0;
            for (uint64_t r : cur) if (keep(col[r])) cur[w++] = r;
            cur.resize(w);
        }
        if (cur.empty()) return true;
        stats[f.table].size = cur.size();
    }
    for (uint64_t a = 0; a < table.size(); a++) {
        unordered_set<uint64_t> &dst = filtered[a];
        if (touched[a]) { dst.reserve(rows[a].size()); dst.insert(rows[a].begin(), rows[a].end()); }
        else {
            const uint64_t n = relations[table[a]].num_tuples;
            dst.reserve(n);
            for (uint64_t r = 0; r < n; r++) dst.insert(r);
        }
    }
    return false;
}

// ------------------------------------------------------------------------------------------------
// intermediates
// ------------------------------------------------------------------------------------------------
// same-alias predicate a.c1 = a.c2 (intermediate.cpp:11-44): a row filter
void parse_table(join_info &join, relList &relation, FilteredRows &filtered, vector<vector<uint64_t> > &intermediate)
{
    const uint64_t *c1 = relation.values[join.column1], *c2 = relation.values[join.column2];
    vector<uint64_t> &col = intermediate[join.table1];
    if (col.empty()) {
        for (uint64_t rowid : filtered.find(join.table1)->second) if (c1[rowid] == c2[rowid]) col.push_back(rowid);
        return;
    }
    // alias already joined: drop the intermediate ROWS that fail the predicate, in every column
    const size_t n = col.size();
    vector<char> ok(n);
    for (size_t e = 0; e < n; e++) ok[e] = c1[col[e]] == c2[col[e]];
    for (vector<uint64_t> &k : intermediate) {
        if (k.empty()) continue;
        size_t w = 0;
        for (size_t e = 0; e < n; e++) if (ok[e]) k[w++] = k[e];
        k.resize(w);
    }
}

template <typename F> static void for_each_pair(const Result &res, F f)
{
    size_t sz = res.size;                                               // head holds `size`, every other page `capacity`
    for (const bucket_info *n = res.head; n; n = n->next) {
        const key_tuple *p = (const key_tuple *)&n[1];
        for (size_t i = 0; i < sz; i++) f(p[i].keyR, p[i].keyS);
        sz = res.capacity;
    }
}

// intermediate.cpp:146-183.  Three cases by which of the two aliases already have a column:
//   neither: the pairs become the two columns;
//   one:     every intermediate row whose rowID of the joined alias equals keyX is extended by its partners;
//   both:    intermediate rows whose (rowID1,rowID2) is a result pair survive.
// The reference finds the matching rows by rescanning the whole intermediate per pair; here an index
// rowID -> row positions is built once.
void update_intermediate(vector<vector<uint64_t> > &intermediate, const Result &results, join_info &join)
{
    const size_t na = intermediate.size();
    vector<vector<uint64_t> > next(na);
    vector<uint64_t> &a1 = intermediate[join.table1], &a2 = intermediate[join.table2];
    if (a1.empty() && a2.empty()) {
        for_each_pair(results, [&](uint64_t r, uint64_t s) { next[join.table1].push_back(r); next[join.table2].push_back(s); });
    } else if (a1.empty() || a2.empty()) {
        const bool first_is_new = a1.empty();
        const vector<uint64_t> &have = first_is_new ? a2 : a1;
        const uint64_t fresh = first_is_new ? join.table1 : join.table2;
        unordered_map<uint64_t, vector<uint32_t> > where;              // rowID of the joined alias -> intermediate rows
        where.reserve(have.size());
        for (size_t e = 0; e < have.size(); e++) where[have[e]].push_back((uint32_t)e);
        vector<size_t> live;
        for (size_t i = 0; i < na; i++) if (!intermediate[i].empty()) live.push_back(i);
        for_each_pair(results, [&](uint64_t r, uint64_t s) {
            const uint64_t known = first_is_new ? s : r, added = first_is_new ? r : s;
            auto it = where.find(known);
            if (it == where.end()) return;
            for (uint32_t e : it->second) {
                for (size_t i : live) next[i].push_back(intermediate[i][e]);
                next[fresh].push_back(added);
            }
        });
    } else {
        struct PairHash { size_t operator()(const std::pair<uint64_t, uint64_t> &p) const { return (size_t)(p.first * 0x9E3779B97F4A7C15ULL ^ p.second); } };
        std::unordered_set<std::pair<uint64_t, uint64_t>, PairHash> hit;
        for_each_pair(results, [&](uint64_t r, uint64_t s) { hit.insert(std::make_pair(r, s)); });
        for (size_t e = 0; e < a1.size(); e++) {
            if (!hit.count(std::make_pair(a1[e], a2[e]))) continue;
            for (size_t i = 0; i < na; i++) if (!intermediate[i].empty()) next[i].push_back(intermediate[i][e]);
        }
    }
    intermediate.swap(next);
}

// ------------------------------------------------------------------------------------------------
// join chain + SUMs (Query.cpp:164-211)
// ------------------------------------------------------------------------------------------------
static std::mutex g_log_mu;

static void log_join(const relation &R, const relation &S, const Result &res)
{
    const char *path = getenv("RHJ_JOIN_LOG");                          // test hook: one line per hot-path call
    if (!path) return;
    size_t m = 0, sz = res.size;
    for (const bucket_info *n = res.head; n; n = n->next) { m += sz; sz = res.capacity; }
    std::lock_guard<std::mutex> lk(g_log_mu);
    FILE *f = fopen(path, "a");
    if (!f) return;
    fprintf(f, "%llu %llu %zu\n", (unsigned long long)R.num_tuples, (unsigned long long)S.num_tuples, m);
    fclose(f);
}

void Query::run_joins(JobScheduler &js, vector<relList> &relations, FilteredRows &filtered)
{
    vector<vector<uint64_t> > intermediate(table.size());
    for (join_info &j : join) {
        if (j.table1 == j.table2) {
            parse_table(j, relations[table[j.table1]], filtered, intermediate);
            if (intermediate[j.table1].empty()) { filtered_out = true; break; }
            continue;
        }
        relation relR, relS;
        relR.create_relation(j.table1, relations[table[j.table1]], j.column1, filtered, intermediate[j.table1]);
        relS.create_relation(j.table2, relations[table[j.table2]], j.column2, filtered, intermediate[j.table2]);
        Result results;
        results.multiRadixHashJoin(js, relR, relS);                     // <-- the hot path (MI355X)
        log_join(relR, relS, results);
        if (results.isEmpty()) { filtered_out = true; break; }
        update_intermediate(intermediate, results, j);
    }
    js.barrier();
    if (filtered_out) return;
    for (proj_info &p : proj) {
        const uint64_t *col = relations[table[p.table]].values[p.column];
        // column_proj (Query.cpp:66-74) over intermediate[p.table]: an alias that is not part of the intermediate
        // (never joined, or dropped when a later join linked two new aliases, intermediate.cpp:147-162) sums to 0,
        // exactly as the reference prints it (tests/golden/edge, generated by the real reference)
        uint64_t sum = 0;
        for (uint64_t rowid : intermediate[p.table]) sum += col[rowid];
        p.sum = sum;
    }
}

// Level-by-level execution of a batch of queries (see rhj_query.h).  Per query the same steps as run_filters / run_joins, the
// join chain advanced one join per round; the joins of a round go to the GPU together.
void Query::execute_batch(JobScheduler &js, vector<Query> &queries, vector<relList> &relations)
{
    struct State {
        Query *q;
        FilteredRows filtered;
        vector<vector<uint64_t> > intermediate;
        size_t next = 0;                                               // next predicate of q->join
        bool done = false;
    };
    vector<State> st(queries.size());
    for (size_t i = 0; i < queries.size(); i++) {
        State &s = st[i];
        s.q = &queries[i];
        s.q->filtered_out = s.q->run_filters(relations, s.filtered);
        s.intermediate.resize(s.q->table.size());
        s.done = s.q->filtered_out;
    }
    for (;;) {
        vector<State *> level;
        for (State &s : st) {
            // same-alias predicates are row filters on the host (parse_table): consume them until a real join comes up
            while (!s.done && s.next < s.q->join.size() && s.q->join[s.next].table1 == s.q->join[s.next].table2) {
                join_info &j = s.q->join[s.next++];
                parse_table(j, relations[s.q->table[j.table1]], s.filtered, s.intermediate);
                if (s.intermediate[j.table1].empty()) { s.q->filtered_out = true; s.done = true; }
            }
            if (!s.done && s.next < s.q->join.size()) level.push_back(&s);
        }
        if (level.empty()) break;
        const size_t n = level.size();
        vector<relation> R(n), S(n);
        vector<Result> res(n);
        vector<relation *> pR(n), pS(n);
        vector<Result *> pres(n);
        for (size_t i = 0; i < n; i++) {
            State &s = *level[i];
            join_info &j = s.q->join[s.next];
            R[i].create_relation(j.table1, relations[s.q->table[j.table1]], j.column1, s.filtered, s.intermediate[j.table1]);
            S[i].create_relation(j.table2, relations[s.q->table[j.table2]], j.column2, s.filtered, s.intermediate[j.table2]);
            pR[i] = &R[i]; pS[i] = &S[i]; pres[i] = &res[i];
        }
        Result::multiRadixHashJoinBatch(js, n, pR.data(), pS.data(), pres.data());     // <-- the hot path, n joins at once
        for (size_t i = 0; i < n; i++) {
            State &s = *level[i];
            log_join(R[i], S[i], res[i]);
            if (res[i].isEmpty()) { s.q->filtered_out = true; s.done = true; continue; }
            update_intermediate(s.intermediate, res[i], s.q->join[s.next]);
            s.next++;
        }
    }
    for (State &s : st) {
        if (s.q->filtered_out) continue;
        for (proj_info &p : s.q->proj) {
            const uint64_t *col = relations[s.q->table[p.table]].values[p.column];
            uint64_t sum = 0;
            for (uint64_t rowid : s.intermediate[p.table]) sum += col[rowid];
            p.sum = sum;
        }
    }
}

void Query::execute(JobScheduler &js, vector<relList> &relations)
{
    // default: the whole query device-resident (rhj_query_dev.cpp); RHJ_QUERY_MODE=host keeps filters and
    // intermediates on the host and sends every equi-join through Result::multiRadixHashJoin with exactly the
    // join inputs the reference builds
    static const bool host_mode = getenv("RHJ_QUERY_MODE") && std::string(getenv("RHJ_QUERY_MODE")) == "host";
    if (!host_mode) { execute_device(js, relations); return; }
    FilteredRows filtered;
    filtered_out = run_filters(relations, filtered);
    if (!filtered_out) run_joins(js, relations, filtered);
}

std::string Query::result_line() const
{
    std::ostringstream os;
    for (size_t i = 0; i < proj.size(); i++) {
        if (i) os << ' ';
        if (filtered_out) os << "NULL"; else os << proj[i].sum;
    }
    return os.str();
}

void Query::print() const { std::cout << result_line() << std::endl; }

// ------------------------------------------------------------------------------------------------
// inter-query threading (MainScheduler.cpp)
// ------------------------------------------------------------------------------------------------
QueryJob::QueryJob(Query &q, vector<relList> &rels) : query(q), relations(rels), js(nullptr) {}
void QueryJob::init(void *arg) { js = (JobScheduler *)arg; }
int QueryJob::run() { query.execute(*js, relations); return 0; }

// start routine of a query thread (MainScheduler.cpp:6-14): a private JobScheduler = a private GPU context
static void *mainThreadWork(void *arg)
{
    MainScheduler *ms = (MainScheduler *)arg;
    JobScheduler mine;
    mine.init(NUM_OF_THREADS);
    ms->threadWork(&mine);
    mine.stop();
    mine.destroy();
    return nullptr;
}

bool MainScheduler::init(size_t n) { return JobScheduler::init(n, mainThreadWork); }
