/*
 * ref_tap.cpp -- link-time tap on the reference's hot-path boundary (test infrastructure only).
 *
 * oracle/Makefile target `_ref/join_tap` compiles the reference's Result.cpp with
 * -DmultiRadixHashJoin=refMultiRadixHashJoin (so the REAL implementation gets a different
 * symbol) and every other reference file unchanged; this file then supplies the symbol the
 * sole caller (Query.cpp:186) binds to, forwards to the real implementation and appends the
 * call's inputs and outputs to $RHJ_TAP_OUT.  No reference source is copied or edited.
 * tests/golden/make_golden.py turns the dump into fixtures.
 */
#include <cstdio>
#include <cstdlib>
#include <mutex>

#define multiRadixHashJoin refMultiRadixHashJoin
#include "Result.h"
#undef multiRadixHashJoin

static std::mutex g_mu;

/* Itanium-ABI name of Result::multiRadixHashJoin(JobScheduler&, relation&, relation&) */
extern "C" void _ZN6Result18multiRadixHashJoinER12JobSchedulerR8relationS3_(Result *self, JobScheduler *js,
                                                                            relation *R, relation *S)
{
    self->refMultiRadixHashJoin(*js, *R, *S);

    size_t count = 0, sz = self->size;
    for (bucket_info *n = self->head; n; n = n->next) { count += sz; sz = self->capacity; }

    const char *path = getenv("RHJ_TAP_OUT");
    if (!path) return;
    std::lock_guard<std::mutex> lk(g_mu);
    FILE *f = fopen(path, "ab");
    if (!f) return;
    uint64_t hdr[4] = {R->num_tuples, S->num_tuples, count, self->head == nullptr ? 1u : 0u};
    fwrite(hdr, sizeof(hdr), 1, f);
    fwrite(R->tuples, sizeof(tuple), R->num_tuples, f);
    fwrite(S->tuples, sizeof(tuple), S->num_tuples, f);
    sz = self->size;
    for (bucket_info *n = self->head; n; n = n->next) { fwrite(&n[1], sizeof(key_tuple), sz, f); sz = self->capacity; }
    fclose(f);
}
