// sharded_host.cpp -- the multi-GPU join driven from a plain C++ host: RCCL (rccl.h) for the two collectives, the C-ABI of
// include/rhj.h for every byte of compute.  One process per GPU:
//
//     sharded_host <rank> <world> <id_file> [rows_per_rank] [zipf]
//
// (rank 0 writes the ncclUniqueId to <id_file>; the other ranks read it -- any launcher that can start `world` processes
// on one node works; the Python driver radixhashjoin_amd/sharded.py runs the same schedule through torch.distributed.)
// The reference has no distributed path; the schedule is SURVEY §8e: rows range-sharded (structs.cpp:146-161
// applied across GPUs), one all-gather of class histograms, one all-to-all of tuples, local radix join, sharded result.
// Verifies its own result: count and order-insensitive checksum of the pair set against the closed form of the PK/FK
// generators, all-reduced over the ranks.
#include <hip/hip_runtime_api.h>
#include <rccl/rccl.h>

#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

#include "../../include/rhj.h"

#define HIPOK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(2); } } while (0)
#define NCCLOK(x) do { ncclResult_t r_ = (x); if (r_ != ncclSuccess) { fprintf(stderr, "%s: %s\n", #x, ncclGetErrorString(r_)); exit(2); } } while (0)
#define RHJOK(ctx, x) do { int r_ = (x); if (r_ != RHJ_OK) { fprintf(stderr, "%s: %s\n", #x, rhj_last_error(ctx)); exit(2); } } while (0)

static const int SHIFT = 20, BITS = 8, C = 1 << BITS;        // owner classes: bits [20, 28) of rhj_mix64(payload)
// tuples per RCCL message at most (512 MiB of payloads).  Test knobs, mirroring sharded.py's max_msg_bytes / force_exchange:
//   RHJ_SHARD_MAX_MSG=<tuples>   smaller messages, so that the piece loop and RCCL's in-order matching of several sends to one
//                                peer run at test sizes
//   RHJ_SHARD_VIA_SELF=1         the rank's OWN segment goes through ncclSend / ncclRecv too, in pieces like a peer's -- the
//                                only way a one-GPU box can execute the send / receive loop at all (world > 1 over RCCL needs
//                                one GPU per rank; unverified on this pool)
static uint64_t env_u64(const char *name, uint64_t dflt)
{
    const char *v = getenv(name);
    return v && *v ? strtoull(v, nullptr, 10) : dflt;
}

// contiguous class ranges of near-equal weight (every rank computes the same cuts from the same gathered histogram)
static std::vector<int> balanced_cuts(const std::vector<uint64_t> &w, int world)
{
    uint64_t total = 0;
    for (uint64_t x : w) total += x;
    std::vector<int> cuts{0};
    uint64_t acc = 0;
    int c = 0;
    for (int r = 1; r < world; r++) {
        const double target = (double)total * r / world;
        while (c < (int)w.size() && (double)acc + (double)w[c] / 2 <= target) acc += w[c++];
        if (c < cuts.back()) c = cuts.back();
        cuts.push_back(c);
    }
    cuts.push_back((int)w.size());
    return cuts;
}

int main(int argc, char **argv)
{
    if (argc < 4) { fprintf(stderr, "usage: %s <rank> <world> <id_file> [rows_per_rank] [zipf]\n", argv[0]); return 1; }
    const int rank = atoi(argv[1]), world = atoi(argv[2]);
    const char *id_file = argv[3];
    const uint64_t n = argc > 4 ? strtoull(argv[4], nullptr, 10) : 4000000;
    const bool zipf = argc > 5 && !strcmp(argv[5], "zipf");
    if (world < 1 || world > 16 || rank < 0 || rank >= world) { fprintf(stderr, "1 <= world <= 16\n"); return 1; }
    const uint64_t MAX_MSG = env_u64("RHJ_SHARD_MAX_MSG", (uint64_t)64 << 20);
    const bool via_self = env_u64("RHJ_SHARD_VIA_SELF", 0) != 0;
    // RHJ_SHARD_PEER=1: no send buffers and no all-to-all -- every rank's class split stores straight into the owners' receive
    // arrays (rhj_shard_split_peer), the peers' HBM mapped with hipIpc handles exchanged by an all-gather.  With one rank the
    // "peer" is the rank's own array (what a one-GPU box can execute); the IPC leg is unverified on this pool.
    const bool peer_mode = env_u64("RHJ_SHARD_PEER", 0) != 0;
    if (MAX_MSG == 0) { fprintf(stderr, "RHJ_SHARD_MAX_MSG must be positive\n"); return 1; }
    uint64_t messages = 0;                                // ncclSend calls issued by this rank
    int ndev = 0;
    HIPOK(hipGetDeviceCount(&ndev));
    const int device = rank % ndev;
    HIPOK(hipSetDevice(device));

    ncclUniqueId id;
    if (rank == 0) {
        NCCLOK(ncclGetUniqueId(&id));
        FILE *f = fopen(id_file, "wb");
        if (!f || fwrite(&id, sizeof id, 1, f) != 1) { perror(id_file); return 2; }
        fclose(f);
    } else {
        for (int tries = 0;; tries++) {
            FILE *f = fopen(id_file, "rb");
            if (f && fread(&id, sizeof id, 1, f) == 1) { fclose(f); break; }
            if (f) fclose(f);
            if (tries > 600) { fprintf(stderr, "no %s\n", id_file); return 2; }
            std::this_thread::sleep_for(std::chrono::milliseconds(100));
        }
    }
    ncclComm_t comm;
    NCCLOK(ncclCommInitRank(&comm, world, id, rank));
    hipStream_t st;
    HIPOK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    rhj_ctx *ctx = nullptr;
    if (rhj_init(device, &ctx) != RHJ_OK) { fprintf(stderr, "%s\n", rhj_last_error(nullptr)); return 2; }
    RHJOK(ctx, rhj_set_stream(ctx, st));                 // engine kernels and RCCL share one stream: ordered by the stream

    // shards of the global PK/FK relations (SURVEY §8d generators), rowIDs global
    const uint64_t nglob = n * world;
    void *dR, *dS;
    RHJOK(ctx, rhj_dev_alloc(ctx, n * 16, &dR));
    RHJOK(ctx, rhj_dev_alloc(ctx, n * 16, &dS));
    RHJOK(ctx, rhj_generate_dev(ctx, 0, (rhj_tuple *)dR, n, rank * n, nglob, 0, 0));
    RHJOK(ctx, rhj_generate_dev(ctx, zipf ? 2 : 1, (rhj_tuple *)dS, n, rank * n, nglob, 42, 900));
    uint64_t exp_cnt = 0, exp_chk = 0;
    RHJOK(ctx, rhj_expected_pkfk_dev(ctx, (const rhj_tuple *)dS, n, &exp_cnt, &exp_chk));

    const auto t0 = std::chrono::steady_clock::now();
    const bool tracing = getenv("RHJ_SHARD_TRACE") != nullptr;                // stage markers on stderr (a hang names its stage)
    auto trace = [&](const char *what) {
        if (tracing) { fprintf(stderr, "[rank %d, %.1f ms] %s\n", rank, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(), what); fflush(stderr); }
    };
    trace("generated");
    // 1. class histograms + rowID ranges
    const int NH = 6;                                   // head words: nR, nS, minR, maxR, minS, maxS
    std::vector<uint64_t> mine(NH + 2 * C);
    mine[0] = n; mine[1] = n;
    RHJOK(ctx, rhj_shard_stats(ctx, 0, (const rhj_tuple *)dR, n, SHIFT, BITS, &mine[NH], &mine[2], &mine[3]));
    RHJOK(ctx, rhj_shard_stats(ctx, 1, (const rhj_tuple *)dS, n, SHIFT, BITS, &mine[NH + C], &mine[4], &mine[5]));
    // 2. ONE all-gather of the count matrix
    void *d_mine, *d_all;
    const size_t words = mine.size();
    RHJOK(ctx, rhj_dev_alloc(ctx, words * 8, &d_mine));
    RHJOK(ctx, rhj_dev_alloc(ctx, words * 8 * world, &d_all));
    RHJOK(ctx, rhj_copy_h2d(ctx, d_mine, mine.data(), words * 8));
    NCCLOK(ncclAllGather(d_mine, d_all, words, ncclUint64, comm, st));
    std::vector<uint64_t> all(words * world);
    RHJOK(ctx, rhj_copy_d2h(ctx, all.data(), d_all, words * 8 * world));   // (synchronises the stream: sizes must be known)
    trace("counts gathered");
    auto H = [&](int r, int rel, int c) { return all[(size_t)r * words + NH + rel * C + c]; };
    std::vector<uint64_t> weight(C, 0);
    for (int r = 0; r < world; r++) for (int c = 0; c < C; c++) weight[c] += H(r, 0, c) + H(r, 1, c);
    const std::vector<int> cuts = balanced_cuts(weight, world);
    uint64_t send[2][16] = {{0}}, recv[2][16] = {{0}}, maxrecv[2] = {0, 0};
    bool small = true;                                   // every rowID below 2^32: they travel as they are
    for (int rel = 0; rel < 2; rel++) {
        for (int d = 0; d < world; d++) {
            for (int c = cuts[d]; c < cuts[d + 1]; c++) send[rel][d] += H(rank, rel, c);
            uint64_t got = 0;
            for (int r = 0; r < world; r++) for (int c = cuts[d]; c < cuts[d + 1]; c++) got += H(r, rel, c);
            if (got > maxrecv[rel]) maxrecv[rel] = got;
        }
        for (int r = 0; r < world; r++) {
            for (int c = cuts[rank]; c < cuts[rank + 1]; c++) recv[rel][r] += H(r, rel, c);
            const uint64_t lo = all[(size_t)r * words + 2 + 2 * rel], hi = all[(size_t)r * words + 3 + 2 * rel];
            if (hi - lo >= (1ull << 32)) { fprintf(stderr, "a shard's rowIDs span 2^32: exchange 16-byte tuples instead (rhj_partition_at)\n"); return 3; }
            if (hi >= (1ull << 32)) small = false;
        }
    }
    rhj_opts plan;
    int mode = rhj_shard_plan(maxrecv[0], maxrecv[1], nullptr, &plan);     // from the LARGEST receive: the same bits on all ranks
    if (mode <= 0 || (mode == RHJ_SHARD_PLAIN && !small)) { fprintf(stderr, "sizes outside the narrow sharded path: exchange 16-byte tuples instead\n"); return 3; }
    if (small) mode = RHJ_SHARD_PLAIN;
    uint64_t row0[2][16] = {{0}};
    for (int rel = 0; rel < 2; rel++) for (int r = 0; r < world; r++) row0[rel][r] = small ? 0 : all[(size_t)r * words + 2 + 2 * rel];

    // 3 + 4. class split into the 12-byte wire format, all-to-all of payloads and rowIDs (R travels while S is split)
    void *sendbuf[2], *rP[2], *rK[2];
    uint64_t m[2], seg[2][17];
    const void *rel_in[2] = {dR, dS};
    if (peer_mode) {
        uint8_t owner[256];
        for (int d = 0; d < world; d++) for (int c = cuts[d]; c < cuts[d + 1]; c++) owner[c] = (uint8_t)d;
        for (int rel = 0; rel < 2; rel++) {
            seg[rel][0] = 0;
            for (int r = 0; r < world; r++) seg[rel][r + 1] = seg[rel][r] + recv[rel][r];
            m[rel] = seg[rel][world];
            RHJOK(ctx, rhj_dev_alloc(ctx, (m[rel] + 2) * 8, &rP[rel]));
            RHJOK(ctx, rhj_dev_alloc(ctx, (m[rel] + 4) * 4, &rK[rel]));
            // where my classes start in every owner's arrays: behind the segments of the senders before me
            uint64_t dst[256];
            for (int d = 0; d < world; d++) {
                uint64_t at = 0;
                for (int r = 0; r < rank; r++) for (int c = cuts[d]; c < cuts[d + 1]; c++) at += H(r, rel, c);
                for (int c = cuts[d]; c < cuts[d + 1]; c++) { dst[c] = at; at += H(rank, rel, c); }
            }
            // the owners' arrays, mapped into this process
            std::vector<void *> pP(world), pK(world);
            pP[rank] = rP[rel];
            pK[rank] = rK[rel];
            if (world > 1) {
                unsigned char mine2[128], *d_h, *d_hall;
                RHJOK(ctx, rhj_ipc_export(ctx, rP[rel], mine2));
                RHJOK(ctx, rhj_ipc_export(ctx, rK[rel], mine2 + 64));
                RHJOK(ctx, rhj_dev_alloc(ctx, 128, (void **)&d_h));
                RHJOK(ctx, rhj_dev_alloc(ctx, 128 * (size_t)world, (void **)&d_hall));
                RHJOK(ctx, rhj_copy_h2d(ctx, d_h, mine2, 128));
                NCCLOK(ncclAllGather(d_h, d_hall, 128, ncclUint8, comm, st));
                std::vector<unsigned char> allh(128 * (size_t)world);
                RHJOK(ctx, rhj_copy_d2h(ctx, allh.data(), d_hall, allh.size()));
                for (int r = 0; r < world; r++) {
                    if (r == rank) continue;
                    RHJOK(ctx, rhj_ipc_open(ctx, &allh[128 * (size_t)r], &pP[r]));
                    RHJOK(ctx, rhj_ipc_open(ctx, &allh[128 * (size_t)r + 64], &pK[r]));
                }
            }
            RHJOK(ctx, rhj_shard_split_peer(ctx, rel, (const rhj_tuple *)rel_in[rel], n, SHIFT, BITS, row0[rel][rank], owner, dst,
                                            pP.data(), pK.data(), world));
            trace("peer split enqueued");
        }
        // every sender's stores into my arrays must have completed before I read them: a one-word all-reduce behind the split
        // kernels of all ranks on the shared stream
        void *d_bar;
        RHJOK(ctx, rhj_dev_alloc(ctx, 64, &d_bar));
        NCCLOK(ncclAllReduce(d_bar, (char *)d_bar + 32, 1, ncclUint64, ncclSum, comm, st));
    }
    for (int rel = 0; rel < 2 && !peer_mode; rel++) {
        RHJOK(ctx, rhj_dev_alloc(ctx, rhj_narrow_bytes(n) + 16, &sendbuf[rel]));
        RHJOK(ctx, rhj_shard_split(ctx, rel, (const rhj_tuple *)rel_in[rel], n, SHIFT, BITS, row0[rel][rank], sendbuf[rel], nullptr));
        seg[rel][0] = 0;
        for (int r = 0; r < world; r++) seg[rel][r + 1] = seg[rel][r] + recv[rel][r];
        m[rel] = seg[rel][world];
        RHJOK(ctx, rhj_dev_alloc(ctx, (m[rel] + 2) * 8, &rP[rel]));
        RHJOK(ctx, rhj_dev_alloc(ctx, (m[rel] + 4) * 4, &rK[rel]));
        const uint64_t *sP = (const uint64_t *)sendbuf[rel];
        const uint32_t *sK = (const uint32_t *)((const char *)sendbuf[rel] + rhj_narrow_key_offset(n));
        // what this rank keeps never enters RCCL: a device copy on the same stream (1 / world of the tuples)
        NCCLOK(ncclGroupStart());
        uint64_t soff = 0;
        for (int d = 0; d < world; d++) {
            if (d == rank && !via_self) {
                HIPOK(hipMemcpyAsync((uint64_t *)rP[rel] + seg[rel][d], sP + soff, send[rel][d] * 8, hipMemcpyDeviceToDevice, st));
                HIPOK(hipMemcpyAsync((uint32_t *)rK[rel] + seg[rel][d], sK + soff, send[rel][d] * 4, hipMemcpyDeviceToDevice, st));
            } else {
                // no message above MAX_MSG tuples (512 MiB of payloads): a segment of a 10^9-row shard is 1 - 4 GB, and this
                // image's RCCL mishandles a single message of 1.6 GB at least to oneself (sharded.py _a2a)
                for (uint64_t o = 0; o < send[rel][d]; o += MAX_MSG) {
                    const uint64_t c = send[rel][d] - o < MAX_MSG ? send[rel][d] - o : MAX_MSG;
                    NCCLOK(ncclSend(sP + soff + o, c, ncclUint64, d, comm, st));
                    NCCLOK(ncclSend(sK + soff + o, c, ncclUint32, d, comm, st));
                    messages += 2;
                }
                for (uint64_t o = 0; o < recv[rel][d]; o += MAX_MSG) {
                    const uint64_t c = recv[rel][d] - o < MAX_MSG ? recv[rel][d] - o : MAX_MSG;
                    NCCLOK(ncclRecv((uint64_t *)rP[rel] + seg[rel][d] + o, c, ncclUint64, d, comm, st));
                    NCCLOK(ncclRecv((uint32_t *)rK[rel] + seg[rel][d] + o, c, ncclUint32, d, comm, st));
                }
            }
            soff += send[rel][d];
        }
        NCCLOK(ncclGroupEnd());
        trace("exchange enqueued");
    }
    // 5 + 6. local fused two-pass partition of what arrived, bucket join (global rowIDs restored as `mode` says)
    for (int rel = 0; rel < 2; rel++)
        RHJOK(ctx, rhj_shard_partition(ctx, rel, (const uint64_t *)rP[rel], (const uint32_t *)rK[rel], m[rel], world, seg[rel], row0[rel], &plan, mode));
    trace("partitions enqueued");
    const uint64_t cap = (m[0] > m[1] ? m[0] : m[1]) + 1024;
    void *d_out;
    RHJOK(ctx, rhj_dev_alloc(ctx, cap * 16, &d_out));
    uint64_t cnt = 0;
    RHJOK(ctx, rhj_shard_join(ctx, (rhj_pair *)d_out, cap, &cnt));
    const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    trace("joined");

    // verification: sum over ranks of (count, checksum) == sum of the local closed forms
    uint64_t chk = 0;
    RHJOK(ctx, rhj_pairs_checksum_dev(ctx, (const rhj_pair *)d_out, cnt, &chk));
    uint64_t v[4] = {cnt, exp_cnt, chk, exp_chk}, tot[4];
    void *d_v;
    RHJOK(ctx, rhj_dev_alloc(ctx, 64, &d_v));
    RHJOK(ctx, rhj_copy_h2d(ctx, d_v, v, 32));
    NCCLOK(ncclAllReduce(d_v, (char *)d_v + 32, 4, ncclUint64, ncclSum, comm, st));
    RHJOK(ctx, rhj_copy_d2h(ctx, tot, (char *)d_v + 32, 32));
    const bool ok = tot[0] == tot[1] && tot[2] == tot[3];
    if (rank == 0)
        printf("{\"world\": %d, \"rows_per_rank\": %llu, \"dist\": \"%s\", \"wire_bytes_per_tuple\": 12, \"rowid_mode\": %d, "
               "\"plan\": [%d, %d, %d], \"pairs_global\": %llu, \"ms_first_join\": %.2f, \"own_segment\": \"%s\", "
               "\"nccl_sends_rank0\": %llu, \"max_tuples_per_message\": %llu, \"verified\": %s}\n",
               world, (unsigned long long)n, zipf ? "zipf0.9" : "uniform", mode, plan.passes, plan.bits1, plan.bits2,
               (unsigned long long)tot[0], ms, peer_mode ? "peer-mapped class split" : via_self ? "ncclSend/ncclRecv to self" : "device copy",
               (unsigned long long)messages, (unsigned long long)MAX_MSG, ok ? "true" : "false");
    rhj_destroy(ctx);
    NCCLOK(ncclCommDestroy(comm));
    return ok ? 0 : 4;
}
