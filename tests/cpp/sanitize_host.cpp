// sanitize_host.cpp -- everything of this repository that runs on the HOST, under AddressSanitizer and
// UndefinedBehaviorSanitizer, in one CPU-only program (tests/test_sanitizers.py builds and runs it; SURVEY.md
// section 5 "sanitizers"; GPU sanitizers are not available on the pool, so this is the CPU build only):
//   (a) the oracle's block restatement and serial spec (oracle/stage1_oracle.c) on the reference's fixtures and on
//       fuzz inputs, in exact-size heap buffers, against each other;
//   (b) the kernel's per-lane math compiled for the host (csrc/lane_math.h via tests/lane_math_host.cpp) against
//       the oracle on the same inputs, strict UTF-8 verdict included;
//   (c) the N-GPU host protocol (csrc/sharded.cpp: msj_shard_speculate / _verify / _global_code,
//       msj_stage1_sharded_submit / _result with its re-run loop, placement, statistics) with world 2..8 ranks as
//       threads, host memory behind msj_sharded_ops, a loopback all-gather and a shard runner that follows the
//       serial spec from a given carry -- every index of every shard, the stitched offsets, the return code and
//       the capacity flag against the serial spec of the whole stream.
// TEST INFRASTRUCTURE: nothing here is part of the product.  The HIP runtime symbols sharded.cpp's default
// operations reference are stubbed (they are never reached: every msj_sharded here has its own operations).
#include <hip/hip_runtime.h>

#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <random>
#include <string>
#include <thread>
#include <deque>
#include <functional>
#include <vector>

#include "../../include/msj_stage1.h"

extern "C" {
int32_t msj_oracle_stage1(const uint8_t *, uint64_t, uint32_t *, uint64_t, uint64_t *);
int32_t msj_oracle_stage1_serial(const uint8_t *, uint64_t, uint32_t *, uint64_t, uint64_t *);
int32_t msj_oracle_utf8(const uint8_t *, uint64_t);
int32_t lane_stage1(const uint8_t *, uint64_t, uint32_t *, uint64_t, uint64_t *, int32_t *);

// ---- stubs: the HIP entry points sharded.cpp names in its DEFAULT operations (never called here)
hipError_t hipMalloc(void **, size_t) { std::abort(); }
hipError_t hipHostMalloc(void **, size_t, unsigned int) { std::abort(); }
hipError_t hipFree(void *) { std::abort(); }
hipError_t hipMemset(void *, int, size_t) { std::abort(); }
hipError_t hipHostFree(void *) { std::abort(); }
hipError_t hipMemcpyAsync(void *, const void *, size_t, hipMemcpyKind, hipStream_t) { std::abort(); }
hipError_t hipStreamSynchronize(hipStream_t) { std::abort(); }
hipError_t hipEventCreate(hipEvent_t *) { std::abort(); }
hipError_t hipEventDestroy(hipEvent_t) { std::abort(); }
hipError_t hipEventRecord(hipEvent_t, hipStream_t) { std::abort(); }
hipError_t hipEventElapsedTime(float *, hipEvent_t, hipEvent_t) { std::abort(); }
hipError_t hipEventSynchronize(hipEvent_t) { std::abort(); }
hipError_t hipEventQuery(hipEvent_t) { std::abort(); }
hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned int) { std::abort(); }
hipError_t hipGetLastError(void) { std::abort(); }
hipError_t hipSetDevice(int) { std::abort(); }
hipError_t hipDeviceGetStreamPriorityRange(int *, int *) { std::abort(); }
hipError_t hipStreamCreateWithPriority(hipStream_t *, unsigned int, int) { std::abort(); }
hipError_t hipStreamDestroy(hipStream_t) { std::abort(); }
int32_t msj_ctx_device(const msj_ctx *) { std::abort(); }
int32_t msj_stage1_shard_device_cv(msj_ctx *, const uint8_t *, uint64_t, uint32_t *, uint64_t, uint32_t, msj_carry *, msj_segment *,
                                   uint32_t, uint32_t *, int32_t, int32_t, int32_t, uint64_t, void *, uint32_t) {
    std::abort();
}
int32_t msj_stage1_shard_device(msj_ctx *, const uint8_t *, uint64_t, uint32_t *, uint64_t, const msj_carry *, msj_carry *,
                                msj_segment *, uint32_t, uint32_t *, int32_t, int32_t, int32_t, uint64_t, void *, uint32_t) {
    std::abort();
}
}

#define CHECK(cond, ...)                                            \
    do {                                                            \
        if (!(cond)) {                                              \
            std::fprintf(stderr, "FAILED %s:%d: %s -- ", __FILE__, __LINE__, #cond); \
            std::fprintf(stderr, __VA_ARGS__);                      \
            std::fprintf(stderr, "\n");                             \
            std::exit(1);                                           \
        }                                                           \
    } while (0)

namespace {

// exact-size heap copies: an access one byte past either end is the sanitizer's to find
struct Bytes {
    uint8_t *p;
    size_t n;
    explicit Bytes(const std::string &s) : p(static_cast<uint8_t *>(std::malloc(s.size() ? s.size() : 1))), n(s.size()) {
        std::memcpy(p, s.data(), s.size());
    }
    ~Bytes() { std::free(p); }
};

std::string soup(std::mt19937_64 &rng, size_t n) {
    static const char a0[] = "\\\\\\\"\"a1 ,:[]{}\n", a4[] = "\"\\\xc3\xa9\xe2\x82\xac\xf0\x9f\x98\x80\xed\xa0\x80\xc0\x01 ,";
    static const std::string alphabets[] = {
        std::string(a0, sizeof a0 - 1), "{}[]:,\"\\ abtrue1.5e\n", "\"xyz\\\" \t:,",
        "{\"k\":\"v w\",\"n\":[1,2.5e3,true,null]} ", std::string(a4, sizeof a4 - 1)};
    const std::string &a = alphabets[rng() % 5];
    std::string s(n, ' ');
    for (auto &c : s) c = a[rng() % a.size()];
    return s;
}

// the serial spec from a given state
struct Serial {
    std::vector<uint32_t> idx;
    uint32_t esc, instr, pnq, bad;
};
Serial serial_run(const uint8_t *d, uint64_t n, uint32_t esc, uint32_t instr, uint32_t pnq) {
    Serial r{{}, esc, instr, pnq, 0};
    for (uint64_t i = 0; i < n; i++) {
        const uint8_t c = d[i];
        const uint32_t escaped = r.esc;
        if (escaped)
            r.esc = 0;
        else if (c == 0x5C)
            r.esc = 1;
        const uint32_t quote = (c == 0x22) && !escaped;
        const uint32_t before = r.instr;
        r.instr ^= quote;
        const bool ws = c == 0x09 || c == 0x0A || c == 0x0D || c == 0x20;
        const bool op = c == 0x0C || c == 0x1A || c == 0x2C || c == 0x3A || c == 0x5B || c == 0x5D || c == 0x7B || c == 0x7D;
        const bool scalar = !(ws || op);
        if ((op || (scalar && !r.pnq)) && !before) r.idx.push_back((uint32_t)i);
        r.bad |= (c <= 0x1F) && r.instr;
        r.pnq = scalar && !quote;
    }
    return r;
}

// ---- (a) + (b)
void oracle_and_lane_math(const std::string &doc, const char *what) {
    Bytes b(doc);
    const uint64_t cap = b.n + 3;
    std::vector<uint32_t *> out;
    uint64_t n[3] = {~0ull, ~0ull, ~0ull};
    int32_t rc[3], u8 = -1;
    for (int k = 0; k < 3; k++) out.push_back(static_cast<uint32_t *>(std::malloc(cap * sizeof(uint32_t))));
    rc[0] = msj_oracle_stage1(b.p, b.n, out[0], cap, &n[0]);
    rc[1] = msj_oracle_stage1_serial(b.p, b.n, out[1], cap, &n[1]);
    rc[2] = lane_stage1(b.p, b.n, out[2], cap, &n[2], &u8);
    CHECK(rc[0] == rc[1] && rc[0] == rc[2], "%s: codes %d %d %d (len %zu)", what, rc[0], rc[1], rc[2], b.n);
    CHECK(n[0] == n[1] && n[0] == n[2], "%s: counts", what);
    if (n[0] != ~0ull) {
        CHECK(std::memcmp(out[0], out[1], (n[0] + 3) * 4) == 0, "%s: block restatement vs serial spec", what);
        CHECK(std::memcmp(out[0], out[2], (n[0] + 3) * 4) == 0, "%s: lane math vs the oracle", what);
    }
    if (b.n) CHECK(u8 == msj_oracle_utf8(b.p, b.n), "%s: utf8 verdict %d", what, u8);
    for (auto p : out) std::free(p);
}

// ---- (c) the protocol, ranks as threads
struct Barrier {
    std::mutex m;
    std::condition_variable cv;
    unsigned n, waiting = 0, gen = 0;
    explicit Barrier(unsigned n_) : n(n_) {}
    void wait() {
        std::unique_lock<std::mutex> lk(m);
        const unsigned g = gen;
        if (++waiting == n) {
            waiting = 0;
            gen++;
            cv.notify_all();
        } else {
            cv.wait(lk, [&] { return gen != g; });
        }
    }
};

struct World {
    unsigned world;
    Barrier bar;
    std::vector<const void *> sends;
    explicit World(unsigned w) : world(w), bar(w), sends(w) {}
};
// An asynchronous fake device (the optional event operations of msj_sharded_ops): a stream is a queue of closures
// that run only when somebody waits -- sync drains a stream, waiting for an event runs its stream up to the record.
// Stream NULL = the submissions' stream, &side_tag = the exchange's.
struct FakeDev {
    unsigned launches = 0;
    std::deque<std::function<void()>> q[2];
    struct Ev {
        bool done = false;
        int stream = -1;
        unsigned gen = 0;
    };
    char side_tag = 0;
    int qi(void *stream) const { return stream == &side_tag ? 1 : 0; }
    void drive(int stream, const Ev *until) {
        while (!q[stream].empty() && !(until && until->done)) {
            auto f = std::move(q[stream].front());
            q[stream].pop_front();
            f();
        }
    }
};

struct RankComm {
    World *w;
    unsigned rank;
    FakeDev *async = nullptr;  // non-null: the collective is enqueued, like ncclAllGather on a stream
};

void gather_now(RankComm *c, const void *d_send, void *d_recv, uint64_t bytes) {
    c->w->sends[c->rank] = d_send;
    c->w->bar.wait();
    for (unsigned g = 0; g < c->w->world; g++) std::memcpy(static_cast<uint8_t *>(d_recv) + g * bytes, c->w->sends[g], bytes);
    c->w->bar.wait();
}

int32_t loop_allgather(void *comm, const void *d_send, void *d_recv, uint64_t bytes, void *stream) {
    RankComm *c = static_cast<RankComm *>(comm);
    if (c->async)
        c->async->q[c->async->qi(stream)].push_back([=] { gather_now(c, d_send, d_recv, bytes); });
    else
        gather_now(c, d_send, d_recv, bytes);
    return MSJ_SUCCESS;
}

int32_t h_alloc(void *, uint64_t bytes, int, void **out) {
    *out = std::malloc(bytes);
    return *out ? MSJ_SUCCESS : MSJ_MEMALLOC;
}
void h_free(void *, void *p, int) { std::free(p); }
int32_t h_copy(void *, void *dst, const void *src, uint64_t bytes, int, void *) {
    std::memcpy(dst, src, bytes);
    return MSJ_SUCCESS;
}
int32_t h_sync(void *, void *) { return MSJ_SUCCESS; }
int32_t h_run(void *user, const uint8_t *d, uint64_t len, uint32_t *idx, uint64_t cap, const msj_carry *cin, msj_carry *cout,
              msj_segment *, uint32_t, int32_t, int32_t is_final, uint64_t trailer_len, void *, uint32_t) {
    (*static_cast<unsigned *>(user))++;
    const Serial r = serial_run(d, len, cin->next_is_escaped, cin->in_string, cin->prev_scalar);
    std::memset(cout, 0, sizeof *cout);
    cout->count = cin->count + r.idx.size();
    cout->bytes = cin->bytes + len;
    cout->in_string = r.instr;
    cout->next_is_escaped = r.esc;
    cout->prev_scalar = r.pnq;
    cout->unescaped_error = r.bad;
    const uint64_t need = r.idx.size() + (is_final ? 3 : 0);
    cout->capacity_error = need > cap;
    for (uint64_t k = 0; k < r.idx.size() && k < cap; k++) idx[k] = r.idx[k];
    if (is_final && need <= cap) {
        idx[r.idx.size()] = idx[r.idx.size() + 1] = (uint32_t)trailer_len;
        idx[r.idx.size() + 2] = 0;
    }
    return MSJ_SUCCESS;
}

// ... and the same operations on the asynchronous fake device
int32_t a_copy(void *user, void *dst, const void *src, uint64_t bytes, int, void *stream) {
    FakeDev *d = static_cast<FakeDev *>(user);
    d->q[d->qi(stream)].push_back([=] { std::memcpy(dst, src, bytes); });
    return MSJ_SUCCESS;
}
int32_t a_sync(void *user, void *stream) {
    FakeDev *d = static_cast<FakeDev *>(user);
    d->drive(d->qi(stream), nullptr);
    return MSJ_SUCCESS;
}
int32_t a_run(void *user, const uint8_t *p, uint64_t len, uint32_t *idx, uint64_t cap, const msj_carry *cin, msj_carry *cout,
              msj_segment *seg, uint32_t ms, int32_t hp, int32_t is_final, uint64_t trailer_len, void *stream, uint32_t flags) {
    FakeDev *d = static_cast<FakeDev *>(user);
    d->q[d->qi(stream)].push_back([=] { (void)h_run(&d->launches, p, len, idx, cap, cin, cout, seg, ms, hp, is_final, trailer_len, stream, flags); });
    return MSJ_SUCCESS;
}
int32_t a_event_create(void *, void **out) {
    *out = new FakeDev::Ev();
    return MSJ_SUCCESS;
}
void a_event_destroy(void *, void *e) { delete static_cast<FakeDev::Ev *>(e); }
int32_t a_event_record(void *user, void *e, void *stream) {
    FakeDev *d = static_cast<FakeDev *>(user);
    FakeDev::Ev *ev = static_cast<FakeDev::Ev *>(e);
    ev->done = false;
    ev->stream = d->qi(stream);
    const unsigned gen = ++ev->gen;
    d->q[ev->stream].push_back([=] {
        if (ev->gen == gen) ev->done = true;
    });
    return MSJ_SUCCESS;
}
int32_t a_event_wait(void *user, void *e) {
    FakeDev *d = static_cast<FakeDev *>(user);
    FakeDev::Ev *ev = static_cast<FakeDev::Ev *>(e);
    if (ev->stream >= 0) d->drive(ev->stream, ev);
    CHECK(ev->done, "waited for an event nothing records");
    return MSJ_SUCCESS;
}
int32_t a_stream_wait(void *user, void *stream, void *e) {
    FakeDev *d = static_cast<FakeDev *>(user);
    d->q[d->qi(stream)].push_back([=] { (void)a_event_wait(user, e); });
    return MSJ_SUCCESS;
}
int32_t a_event_query(void *, void *e) { return static_cast<FakeDev::Ev *>(e)->done ? 1 : 0; }

struct RankOut {
    int32_t code = -99;
    uint64_t total = 0;
    msj_shard_placement place{};
    std::vector<uint32_t> idx;
    unsigned launches = 0;
    uint64_t reruns = 0;
};

uint64_t g_reruns = 0, g_clipped = 0, g_async = 0, g_async_reruns = 0;

void protocol_case(std::mt19937_64 &rng, const std::string &doc, const std::vector<uint64_t> &cuts, int short_rank) {
    const unsigned world = (unsigned)cuts.size() - 1;
    World w(world);
    std::vector<RankOut> out(world);
    std::vector<std::thread> th;
    const bool given = rng() & 1;  // the speculation: handed in by the caller, or derived by the library from the bytes
    const bool async = rng() & 2;  // the operations: complete before they return, or the asynchronous fake device
    for (unsigned rank = 0; rank < world; rank++)
        th.emplace_back([&, rank] {
            FakeDev dev;
            RankComm comm{&w, rank, async ? &dev : nullptr};
            msj_exchange x{&comm, loop_allgather, rank, world, 0, 0};
            unsigned launches = 0;
            msj_sharded_ops ops{&launches, h_alloc, h_free, h_copy, h_sync, h_run};  // no events: synchronous
            if (async)
                ops = msj_sharded_ops{&dev,           h_alloc,        h_free,       a_copy,        a_sync,
                                      a_run,          a_event_create, a_event_destroy, a_event_record, a_event_wait,
                                      a_stream_wait,  a_event_query,  nullptr,      &dev.side_tag};
            msj_sharded *sh = nullptr;
            CHECK(msj_sharded_create(nullptr, &x, &ops, &sh) == MSJ_SUCCESS, "create");
            const uint64_t lo = cuts[rank], hi = cuts[rank + 1], len = hi - lo;
            // the "device" buffer: exactly the 64-byte halo (ranks > 0) and the shard
            const uint64_t halo = rank ? 64 : 0;
            std::string pad(rank && lo < 64 ? 64 - lo : 0, '\0');
            Bytes buf(pad + doc.substr(lo - (halo - pad.size()), halo - pad.size() + len));
            const uint64_t cap = (int)rank == short_rank ? len / 3 : len + 3;
            uint32_t *idx = static_cast<uint32_t *>(std::malloc((cap ? cap : 1) * sizeof(uint32_t)));
            msj_carry spec;
            if (given && rank) CHECK(msj_shard_speculate(buf.p + pad.size(), 64 - pad.size(), buf.p + 64, len < 4096 ? len : 4096, &spec) == 0, "speculate");
            // twice: slots are reused.  Asynchronous device: both submissions in flight before the first result
            uint32_t tickets[2] = {99, 99};
            auto submit = [&](int round) {
                const int32_t src = msj_stage1_sharded_submit(sh, buf.p + halo, len, idx, cap, doc.size(), rank > 0,
                                                              (given && rank) ? &spec : nullptr, nullptr, 0, nullptr, 0, &tickets[round]);
                CHECK(src == MSJ_SUCCESS, "submit: %d (rank %u, len %llu)", src, rank, (unsigned long long)len);
            };
            auto result = [&](int round) {
                msj_carry local, used;
                CHECK(msj_stage1_sharded_result(sh, tickets[round], &out[rank].code, &out[rank].total, &local, &used, &out[rank].place) == MSJ_SUCCESS,
                      "result");
                CHECK(local.count == out[rank].place.count && out[rank].place.bytes == len, "placement counts");
            };
            if (async) {
                submit(0);
                submit(1);
                CHECK(given && rank ? msj_sharded_ticket_state(sh, tickets[1]) == 0 : true, "nothing has run before anybody waits");
                result(0);
                result(1);
                launches = dev.launches;
            } else {
                for (int round = 0; round < 2; round++) {
                    submit(round);
                    result(round);
                }
            }
            out[rank].idx.assign(idx, idx + (out[rank].place.count < cap ? out[rank].place.count : cap));
            out[rank].launches = launches;
            out[rank].reruns = msj_sharded_reruns(sh);
            msj_sharded_stats st;
            CHECK(msj_sharded_get_stats(sh, &st) == MSJ_SUCCESS && st.results == 2 && st.reruns == out[rank].reruns, "stats");
            msj_sharded_destroy(sh);
            std::free(idx);
        });
    for (auto &t : th) t.join();
    const Serial all = serial_run(reinterpret_cast<const uint8_t *>(doc.data()), doc.size(), 0, 0, 0);
    int32_t want = all.instr ? MSJ_UNCLOSED_STRING : all.bad ? MSJ_UNESCAPED_CHARS : all.idx.empty() ? MSJ_EMPTY : MSJ_SUCCESS;
    uint64_t begin = 0;
    bool clipped = false;
    for (unsigned rank = 0; rank < world; rank++) {
        const RankOut &r = out[rank];
        CHECK(r.total == all.idx.size(), "total %llu != %zu", (unsigned long long)r.total, all.idx.size());
        CHECK(r.place.index_begin == begin && r.place.byte_base == cuts[rank], "rank %u: placement (%llu, %llu), want (%llu, %llu)", rank,
              (unsigned long long)r.place.index_begin, (unsigned long long)r.place.byte_base, (unsigned long long)begin,
              (unsigned long long)cuts[rank]);
        for (size_t k = 0; k < r.idx.size(); k++)
            CHECK(r.idx[k] + cuts[rank] == all.idx[begin + k], "rank %u index %zu", rank, k);
        const uint64_t need = r.place.count + (rank + 1 == world ? 3 : 0);
        clipped |= (int)rank == short_rank && need > (cuts[rank + 1] - cuts[rank]) / 3;
        begin += r.place.count;
    }
    CHECK(begin == all.idx.size(), "counts add up");
    for (const RankOut &r : out) g_reruns += r.reruns;
    if (async) {
        g_async++;
        for (const RankOut &r : out) g_async_reruns += r.reruns;
    }
    g_clipped += clipped;
    if (want == MSJ_SUCCESS || want == MSJ_EMPTY) want = clipped ? MSJ_CAPACITY : want;
    for (unsigned rank = 0; rank < world; rank++) CHECK(out[rank].code == want, "rank %u: code %d, want %d", rank, out[rank].code, want);
}

}  // namespace

int main(int argc, char **argv) {
    const uint64_t seed = argc > 1 ? std::strtoull(argv[1], nullptr, 10) : 1;
    const int rounds = argc > 2 ? std::atoi(argv[2]) : 300;
    std::mt19937_64 rng(seed);
    // the reference's fixtures (first line of each file), handed in by the test as further arguments
    for (int k = 3; k < argc; k++) {
        std::FILE *f = std::fopen(argv[k], "rb");
        CHECK(f, "cannot open %s", argv[k]);
        std::string line;
        for (int c; (c = std::fgetc(f)) != EOF && c != '\n';) line.push_back((char)c);
        std::fclose(f);
        oracle_and_lane_math(line, argv[k]);
    }
    size_t bytes = 0;
    for (int i = 0; i < rounds * 10; i++) {
        static const size_t sizes[] = {0, 1, 2, 63, 64, 65, 127, 128, 129, 191, 192, 255, 256, 257, 1000, 4097};
        const size_t n = (i % 3 == 0) ? sizes[(i / 3) % 16] : rng() % 700;
        const std::string d = soup(rng, n);
        oracle_and_lane_math(d, "fuzz");
        bytes += n;
    }
    // msj_shard_speculate on exact-size halo / head buffers of every length
    for (int i = 0; i < rounds * 10; i++) {
        Bytes halo(soup(rng, rng() % 65)), head(soup(rng, (rng() & 3) ? rng() % 80 : rng() % 5000));
        msj_carry c;
        CHECK(msj_shard_speculate(halo.p, halo.n, head.p, head.n, &c) == MSJ_SUCCESS, "speculate");
        CHECK(c.in_string <= 1 && c.next_is_escaped <= 1 && c.prev_scalar <= 1, "speculate: bits");
    }
    int cases = 0;
    for (int i = 0; i < rounds; i++) {
        const unsigned world = 2 + rng() % 7;
        const size_t n = world * 70 + rng() % 3000;
        const std::string d = soup(rng, n);
        std::vector<uint64_t> cuts{0};
        const size_t base = n / world;  // >= 70: every shard (and every halo) has at least 67 bytes
        for (unsigned g = 1; g < world; g++) cuts.push_back(g * base + rng() % (base - 66));
        cuts.push_back(n);
        protocol_case(rng, d, cuts, (i % 5 == 4) ? (int)(rng() % world) : -1);
        cases++;
    }
    CHECK(g_reruns > 0 && g_clipped > 0, "the re-run loop and the capacity flag were exercised (%llu, %llu)", (unsigned long long)g_reruns,
          (unsigned long long)g_clipped);
    CHECK(g_async > 0 && g_async_reruns > 0, "the event operations (two submissions in flight) and re-runs under them were exercised (%llu, %llu)",
          (unsigned long long)g_async, (unsigned long long)g_async_reruns);
    {  // shards whose first 4 KiB (and first 64 KiB) decide nothing: the library reads a longer head, never past the shard
        std::string d = "[\"";
        for (int i = 0; i < 30000; i++) d += "1 2 3 ";
        d += "x\",7,\"";
        for (int i = 0; i < 3000; i++) d += "4 5 ";
        d += "\"]";
        const uint64_t n = d.size();
        const uint64_t before = g_reruns;
        protocol_case(rng, d, {0, 80, n}, -1);
        protocol_case(rng, d, {0, 80, n - 5000, n}, -1);       // the last shard starts inside the second string, 5 000 bytes long
        protocol_case(rng, d, {0, n - 70000, n - 66, n}, -1);  // a 66-byte shard behind a 70 KB one
        (void)before;
        cases += 3;
    }
    std::printf("sanitize_host ok: seed %llu, %d fuzz documents (%zu bytes) through the oracle and the lane math, %d sharded streams "
                "(%llu refuted speculations repaired, %llu with an index buffer too small)\n",
                (unsigned long long)seed, rounds * 10, bytes, cases, (unsigned long long)g_reruns, (unsigned long long)g_clipped);
    return 0;
}
