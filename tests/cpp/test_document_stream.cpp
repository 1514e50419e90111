// C++ check of include/document_stream.hpp (multi-document mode above the C ABI): a stream of newline-separated
// documents of known offsets goes through DocumentStream at several window sizes; the documents found must be
// the ones written, the tokens of every window must equal the whole stream's, and the error cases must give the
// reference's codes.  Run on the GPU box by tests/test_documents.py.
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

#include "document_stream.hpp"

using namespace mojo_simdjson;

#define CHECK(cond)                                                                  \
    do {                                                                             \
        if (!(cond)) {                                                               \
            std::fprintf(stderr, "%s:%d: check failed: %s\n", __FILE__, __LINE__, #cond); \
            std::exit(1);                                                            \
        }                                                                            \
    } while (0)

static uint64_t rng_state = 0x9E3779B97F4A7C15ull;
static uint32_t rnd(uint32_t n) {
    rng_state = rng_state * 6364136223846793005ull + 1442695040888963407ull;
    return (uint32_t)(rng_state >> 33) % n;
}

static std::string document(int i) {
    switch (rnd(5)) {
        case 0: return std::to_string(i * 7 - 3);
        case 1: return "\"s" + std::string(rnd(40), 'x') + "\\\" [not] {a} bracket\"";
        case 2: return "[" + std::to_string(i) + ",[1,2,{\"k\":null}],\"" + std::string(rnd(100), 'y') + "\"]";
        case 3: return "{\"id\":" + std::to_string(i) + ",\"u\":{\"n\":\"caf\xc3\xa9 \xe4\xb8\xad\",\"t\":[true,false]},\"e\":1.5e-3}";
        default: return "true";
    }
}

int main() {
    msj_ctx *ctx = nullptr;
    CHECK(msj_ctx_create(0, &ctx) == 0);
    std::string stream;
    std::vector<uint64_t> starts;
    const char *seps[] = {"\n", " ", "\r\n", "  \n"};
    for (int i = 0; i < 20000; i++) {
        if (i) stream += seps[rnd(4)];
        starts.push_back(stream.size());
        stream += document(i);
    }
    uint8_t *d_buf = nullptr;
    CHECK(msj_device_alloc(ctx, stream.size(), reinterpret_cast<void **>(&d_buf)) == 0);
    CHECK(msj_copy_to_device(ctx, d_buf, stream.data(), stream.size(), nullptr) == 0);

    std::vector<uint64_t> all_tokens;  // of the first pass, compared with the others
    for (uint64_t window : {1024ull, 4096ull, 65536ull, 1ull << 20, 1ull << 28}) {
        DocumentStream ds(ctx, d_buf, stream.size(), window);
        CHECK(ds.error() == 0);
        DocumentWindow w;
        std::vector<uint64_t> offsets, tokens;
        while (ds.next(w)) {
            CHECK(w.base % 16 == 0 && w.n_documents > 0);
            std::vector<uint32_t> idx(w.n_tokens), first(w.n_documents);
            CHECK(msj_copy_to_host(ctx, idx.data(), w.d_idx, idx.size() * 4, nullptr) == 0);
            CHECK(msj_copy_to_host(ctx, first.data(), w.d_doc_first, first.size() * 4, nullptr) == 0);
            for (uint32_t t : first) offsets.push_back(w.base + idx[t]);
            for (uint32_t o : idx) tokens.push_back(w.base + o);
        }
        CHECK(ds.error() == 0);
        CHECK(offsets == starts);
        if (all_tokens.empty()) all_tokens = tokens;
        CHECK(tokens == all_tokens);
        std::printf("window %llu: %llu windows, %zu documents, %zu tokens\n", (unsigned long long)window,
                    (unsigned long long)ds.windows(), offsets.size(), tokens.size());
    }
    {   // a document larger than the window
        DocumentStream ds(ctx, d_buf, stream.size(), 64);
        DocumentWindow w;
        while (ds.next(w)) {}
        CHECK(ds.error() == errors::CAPACITY);
    }
    struct { const char *tail; int code; } bad[] = {{" {\"a\":[1,2", errors::TAPE_ERROR}, {" {\"a\":\"xy", errors::UNCLOSED_STRING}, {" ] 1", errors::TAPE_ERROR}};
    for (auto &b : bad) {
        const std::string s = stream + b.tail;
        uint8_t *d = nullptr;
        CHECK(msj_device_alloc(ctx, s.size(), reinterpret_cast<void **>(&d)) == 0);
        CHECK(msj_copy_to_device(ctx, d, s.data(), s.size(), nullptr) == 0);
        DocumentStream ds(ctx, d, s.size(), 65536);
        DocumentWindow w;
        while (ds.next(w)) {}
        CHECK(ds.error() == b.code);
        CHECK(msj_device_free(ctx, d) == 0);
    }
    CHECK(msj_device_free(ctx, d_buf) == 0);
    msj_ctx_destroy(ctx);
    std::printf("test_document_stream ok\n");
    return 0;
}
