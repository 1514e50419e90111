// Multi-document mode above the C ABI of include/msj_stage1.h, in C++.
//
// The reference parses one document per call and marks streaming as to do
// (src/mojo_simdjson/generic/stage2/tape_builder.mojo:25 "TODO: add streaming";
// generic/stage1/json_structural_indexer.mojo:153,169 are where upstream simdjson's streaming_partial /
// streaming_final steps were left out).  This is the role of upstream's `document_stream` (parse_many): cut a
// stream of concatenated documents into windows, index a window, find where its last complete document ends,
// start the next window there -- with the three device passes of msj_stage1_shard_device(is_final = 0),
// msj_tokens_device and msj_documents_device per window.  Same logic as mojo_simdjson_amd/document_stream.py.
//
// Offsets in Window::d_idx are relative to Window::base; the device arrays of a window are reused by the next.
#pragma once
#include <cstdint>
#include <string>

#include "dom_parser_implementation.hpp"
#include "msj_stage1.h"

namespace mojo_simdjson {

struct DocumentWindow {
    uint64_t base = 0;         // byte offset of the window in the stream (16-byte aligned)
    uint64_t length = 0;       // bytes indexed from there
    uint64_t consumed = 0;     // bytes that belong to its complete documents: the next window starts at base + consumed
    uint64_t n_tokens = 0;     // structurals of the complete documents
    uint64_t n_documents = 0;  // complete documents
    bool utf8_error = false;   // stage 1's verdict for the window
    const uint32_t *d_idx = nullptr;        // device: n_tokens offsets relative to base
    const uint8_t *d_type = nullptr;        // device: n_tokens type bytes
    const int32_t *d_depth = nullptr;       // device: n_tokens depths
    const uint32_t *d_doc_first = nullptr;  // device: n_documents token indices
};

class DocumentStream {
  public:
    // d_buf: the stream in device memory (16-byte aligned), len bytes.  window: bytes indexed per step, a multiple
    // of 16 (a document must fit in one window, like upstream's batch_size); index_capacity: structurals a window
    // may hold (0: one per byte up to 64 MiB windows, one per two bytes beyond).
    DocumentStream(msj_ctx *ctx, const uint8_t *d_buf, uint64_t len, uint64_t window = 1ull << 28, uint64_t index_capacity = 0,
                   uint32_t flags = 0)
        : ctx_(ctx), d_buf_(d_buf), len_(len), window_(window < (1ull << 31) ? window : (1ull << 31)), flags_(flags & 3u) {
        const uint64_t w = (window_ + 16 < (len_ > 16 ? len_ : 16)) ? window_ + 16 : (len_ > 16 ? len_ : 16);
        capacity_ = index_capacity ? index_capacity : (w <= (64ull << 20) ? w + 3 : w / 2 + 1024);
        bool ok = window_ >= 64 && window_ % 16 == 0 && (reinterpret_cast<uintptr_t>(d_buf) & 15u) == 0;
        ok = ok && msj_device_alloc(ctx_, capacity_ * 4, reinterpret_cast<void **>(&idx_)) == 0;
        ok = ok && msj_device_alloc(ctx_, capacity_, reinterpret_cast<void **>(&type_)) == 0;
        ok = ok && msj_device_alloc(ctx_, capacity_ * 4, reinterpret_cast<void **>(&depth_)) == 0;
        ok = ok && msj_device_alloc(ctx_, capacity_ * 4, reinterpret_cast<void **>(&first_)) == 0;
        ok = ok && msj_device_alloc(ctx_, 256, reinterpret_cast<void **>(&small_)) == 0;
        if (ok) {
            const msj_carry zero{};
            ok = msj_copy_to_device(ctx_, small_, &zero, sizeof zero, nullptr) == 0;  // small_[0..64): carry in
        }
        error_ = ok ? errors::SUCCESS : errors::MEMALLOC;
    }
    ~DocumentStream() {
        msj_device_free(ctx_, idx_);
        msj_device_free(ctx_, type_);
        msj_device_free(ctx_, depth_);
        msj_device_free(ctx_, first_);
        msj_device_free(ctx_, small_);
    }
    DocumentStream(const DocumentStream &) = delete;
    DocumentStream &operator=(const DocumentStream &) = delete;

    bool done() const { return pos_ >= len_ || error_ != errors::SUCCESS; }
    ErrorType error() const { return error_; }          // reference codes (errors.mojo): 0, CAPACITY, TAPE_ERROR, 14, 15 ...
    const std::string &message() const { return message_; }
    uint64_t windows() const { return windows_; }

    // Index the next window.  Returns false at the end of the stream or on an error (see error()).
    bool next(DocumentWindow &out) {
        if (done()) return false;
        const uint64_t base = pos_ & ~15ull, skip = pos_ - base;
        const uint64_t wlen = (window_ + skip < len_ - base) ? window_ + skip : len_ - base;
        const bool last = base + wlen == len_;
        msj_carry *d_cin = reinterpret_cast<msj_carry *>(small_), *d_cout = d_cin + 1;
        msj_tokens_result *d_tok = reinterpret_cast<msj_tokens_result *>(small_ + 128);
        msj_documents_result *d_doc = reinterpret_cast<msj_documents_result *>(small_ + 192);
        uint32_t nseg = 0;
        // a window is a non-final shard with zero carries: no return code, no trailer, an unclosed string or a cut
        // UTF-8 character at its end is not an error (the next window starts in front of it)
        int32_t rc = msj_stage1_shard_device(ctx_, d_buf_ + base, wlen, idx_, capacity_, d_cin, d_cout, nullptr, 0, &nseg, 0, 0, 0, 0,
                                             nullptr, flags_ | MSJ_FLAG_SKIP(skip));
        msj_carry carry{};
        if (rc == 0) rc = msj_carry_fetch(ctx_, d_cout, &carry, nullptr);
        if (rc != 0) return fail(rc > 0 ? rc : errors::UNEXPECTED_ERROR, "stage 1 failed");
        if (carry.internal_error) return fail(errors::CAPACITY, "more structurals in a window than index_capacity");
        const uint64_t n = carry.count;
        rc = msj_tokens_device(ctx_, d_buf_ + base, wlen, idx_, n, type_, depth_, nullptr, d_tok, nullptr);
        if (rc == 0) rc = msj_documents_device(ctx_, d_buf_ + base, wlen, (last ? MSJ_DOCS_FINAL : 0) | MSJ_DOCS_AFTER_TOKENS, idx_, n, type_, depth_, d_cout, first_, capacity_, d_doc, nullptr);
        struct {  // one read for both result structs: small_[128 .. 224)
            msj_tokens_result tok;
            uint8_t pad[64 - sizeof(msj_tokens_result)];
            msj_documents_result doc;
        } both{};
        if (rc == 0) rc = msj_copy_to_host(ctx_, &both, d_tok, sizeof both, nullptr);
        if (rc != 0) return fail(rc > 0 ? rc : errors::UNEXPECTED_ERROR, "token pre-pass failed");
        const msj_tokens_result &tok = both.tok;
        const msj_documents_result &doc = both.doc;
        const bool cut = doc.n_complete < doc.n_documents;
        if (carry.unescaped_error) return fail(errors::UNESCAPED_CHARS, "control character inside a string");
        if (tok.min_depth < 0) return fail(errors::TAPE_ERROR, "closing bracket without an opening one");
        if (last && cut) return fail(carry.in_string ? errors::UNCLOSED_STRING : errors::TAPE_ERROR, "the stream ends inside a document");
        if (cut && doc.n_complete == 0) return fail(errors::CAPACITY, "a document does not fit in one window");
        windows_++;
        out.base = base;
        out.length = wlen;
        out.consumed = cut ? doc.resume_offset : wlen;
        out.n_tokens = doc.tokens_complete;
        out.n_documents = doc.n_complete;
        out.utf8_error = carry.utf8_error != 0;
        out.d_idx = idx_;
        out.d_type = type_;
        out.d_depth = depth_;
        out.d_doc_first = first_;
        pos_ = base + out.consumed;
        return true;
    }

  private:
    bool fail(ErrorType code, const char *what) {
        error_ = code;
        message_ = std::string(what) + " (window at " + std::to_string(pos_ & ~15ull) + ")";
        return false;
    }
    msj_ctx *ctx_;
    const uint8_t *d_buf_;
    uint64_t len_, window_, capacity_ = 0, pos_ = 0, windows_ = 0;
    uint32_t flags_;
    uint32_t *idx_ = nullptr, *first_ = nullptr;
    uint8_t *type_ = nullptr, *small_ = nullptr;
    int32_t *depth_ = nullptr;
    ErrorType error_ = errors::SUCCESS;
    std::string message_;
};

}  // namespace mojo_simdjson
