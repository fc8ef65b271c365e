/*
 * rhj_query.h -- the callers and data formats on either side of the hot path (SURVEY §8f "next"):
 * relList (column files), Query (filters -> join chain -> SUM projections), intermediate results,
 * MainScheduler/QueryJob and the `join` CLI protocol.  Same type names, members and signatures as
 * the reference's structs.h:11-31, Query.h, intermediate.h, MainScheduler.h so that the reference's
 * join.cpp compiles against it; the implementation (rhj_query.cpp) is new:
 *   - every equi-join goes through Result::multiRadixHashJoin -> rhj_join (MI355X), with exactly the
 *     inputs the reference would build (filtered rowIDs, or DISTINCT rowIDs of an alias already in the
 *     intermediate), so the 94 joins of small.work are the same 94 joins;
 *   - update_intermediate is position-indexed (O(|result| + |intermediate|)) instead of the
 *     reference's O(|result| x |intermediate|) rescans (intermediate.cpp:52-87), which account for
 *     ~99 % of its small.work wall time.  Row ORDER of intermediates differs; every
 *     consumer is order-insensitive (SUMs, de-duplication, value matching).
 */
#ifndef RHJ_QUERY_H
#define RHJ_QUERY_H

#include <string>
#include <unordered_map>
#include <unordered_set>
#include <vector>

#include "rhj_compat.h"

struct relList {                              /* structs.h:11-22 */
    uint64_t num_tuples;
    uint64_t num_columns;
    uint64_t **values;                        /* values[column][row], column-major, mmap'd */
    uint64_t *col_min;
    uint64_t *col_max;
    uint64_t *distinct;
    explicit relList(char *filename);         /* file = [num_tuples, num_columns, column 0 ..., column 1 ...] as uint64 */
    void destroy();
};

struct relList_stats {                        /* structs.h:24-31 */
    uint64_t size;
    uint64_t *low;
    uint64_t *max;
    uint64_t *distinct;
    void fill(relList &relation);
};

struct join_info {                            /* Query.h:8-15 */
    join_info(uint64_t table1, uint64_t column1, uint64_t table2, uint64_t column2);
    uint64_t table1, column1, table2, column2;
};

struct filter_info {                          /* Query.h:17-24 */
    filter_info(uint64_t table, uint64_t column, int op, uint64_t number);
    uint64_t table, column;
    int op;                                   /* '<', '>' or '=' */
    uint64_t number;
};

struct proj_info {                            /* Query.h:26-32 */
    proj_info(uint64_t table, uint64_t column);
    uint64_t table, column, sum;
};

struct Query {                                /* Query.h:34-59 */
    std::vector<uint64_t> table;
    std::vector<join_info> join;
    std::vector<filter_info> filter;
    std::vector<proj_info> proj;
    std::vector<relList_stats> stats;
    bool filtered_out;

    explicit Query(int ch);                   /* ch = first character of the query line on stdin */
    explicit Query(const std::string &line);  /* "tables|predicates|projections" */
    bool read_relations(int ch);
    void read_predicates();
    void read_projections();
    void execute(JobScheduler &js, std::vector<relList> &relations);
    /* the same query executed device-resident (rhj_query_dev.cpp): what execute() does unless
       $RHJ_QUERY_MODE == "host" */
    void execute_device(JobScheduler &js, std::vector<relList> &relations);
    bool run_filters(std::vector<relList> &relations,
                     std::unordered_map<uint64_t, std::unordered_set<uint64_t> > &filtered);
    void run_joins(JobScheduler &js, std::vector<relList> &relations,
                   std::unordered_map<uint64_t, std::unordered_set<uint64_t> > &filtered);
    void print() const;
    std::string result_line() const;
    /* NOT in the reference: a batch of queries executed LEVEL BY LEVEL from one thread (host mode semantics): all filters,
       then the first join of every query in ONE Result::multiRadixHashJoinBatch call, the intermediates, the second joins ...
       -- the joins of one level are independent, and sixteen of them share a GPU launch (rhj_join_batch).  Same results as
       execute() per query.  (join_main: RHJ_QUERY_MODE=batch.) */
    static void execute_batch(JobScheduler &js, std::vector<Query> &queries, std::vector<relList> &relations);
private:
    std::string text_;
    size_t pos_ = 0;
    void parse_all();
};

/* intermediate.h:10-14 */
void parse_table(join_info &join, relList &relation,
                 std::unordered_map<uint64_t, std::unordered_set<uint64_t> > &filtered,
                 std::vector<std::vector<uint64_t> > &intermediate);
void update_intermediate(std::vector<std::vector<uint64_t> > &intermediate, const Result &results, join_info &join);

class QueryJob : public Job {                 /* MainScheduler.h:9-19 */
    Query &query;
    std::vector<relList> &relations;
    JobScheduler *js;
public:
    QueryJob(Query &query, std::vector<relList> &relations);
    void init(void *js) override;
    int run() override;
};

class MainScheduler : public JobScheduler {   /* MainScheduler.h:24-27 */
public:
    bool init(size_t num_of_threads) override;
};

#endif /* RHJ_QUERY_H */
