// dom_parser_implementation.hpp -- C++ mirror of the reference's host-side facade for the stage-1 path,
// above the C ABI of include/msj_stage1.h (the reference is compiled Mojo; this image has no Mojo
// toolchain, so the host side that a Mojo maintainer would write as the shim in INTEGRATION.md is
// mirrored here in C++ and, for the Python tests, in mojo_simdjson_amd/dom_parser_implementation.py).
//
// Mirrors struct DomParserImplementation of
//   src/mojo_simdjson/include/generic/dom_parser_implementation.mojo:20-95
// with the same member names and the same behaviour on this path:
//   stage1(buffer) -> ErrorType   (:59-69)  allocate(len(buffer)); buf / length recorded; the callee fills
//                                            structural_indexes[0..n), the three trailer words,
//                                            n_structural_indexes, next_structural_index = 0
//   allocate(amount)              (:85-89)  structural_indexes resized to `amount` (+3: the reference's
//                                            own trailer write overruns `amount` when every byte is
//                                            structural, include/msj_stage1.h "Capacity")
//   capacity(), max_depth()       (:53-57)
// Error codes: mojo_simdjson::errors, src/mojo_simdjson/errors.mojo:2-36.
// HIP only: stage1() returns MSJ_ERR_NO_DEVICE (-2) without a usable GPU; there is no CPU path here.
#pragma once
#include <cstddef>
#include <cstdint>
#include <string>
#include <string_view>
#include <vector>

#include "msj_stage1.h"

namespace mojo_simdjson {

using ErrorType = int;

namespace errors {  // src/mojo_simdjson/errors.mojo:2-26
constexpr ErrorType SUCCESS = 0;
constexpr ErrorType CAPACITY = 1;
constexpr ErrorType MEMALLOC = 2;
constexpr ErrorType TAPE_ERROR = 3;
constexpr ErrorType DEPTH_ERROR = 4;
constexpr ErrorType UTF8_ERROR = 11;
constexpr ErrorType EMPTY = 13;
constexpr ErrorType UNESCAPED_CHARS = 14;
constexpr ErrorType UNCLOSED_STRING = 15;
constexpr ErrorType UNEXPECTED_ERROR = 24;
}  // namespace errors

struct DomParserImplementation {
    const uint8_t *buf = nullptr;                 // :23 (borrowed, the caller keeps it alive through stage 2)
    size_t length = 0;                            // :24
    uint32_t n_structural_indexes = 0;            // :26
    std::vector<uint32_t> structural_indexes;     // :27
    uint32_t next_structural_index = 0;           // :28
    int utf8_verdict = 0;                         // not in the reference (its checker is a stub): 0 valid, 11 invalid
    uint32_t flags = 0;                           // MSJ_FLAG_* for the calls below

    size_t capacity() const { return _capacity; }  // :56-57
    int max_depth() const { return _max_depth; }   // :53-54

    ErrorType stage1(const std::string &buffer) { return stage1(std::string_view(buffer)); }  // :59-60
    ErrorType stage1(std::string_view buffer) {                                               // :62-63
        return stage1(reinterpret_cast<const uint8_t *>(buffer.data()), buffer.size());
    }
    ErrorType stage1(const uint8_t *buffer, size_t len) {  // :65-69 (Span[UInt8])
        allocate(len);
        buf = buffer;
        length = len;
        uint64_t n = 0;
        int32_t verdict = 0;
        // replaces `return JsonStructuralIndexer.index[128](buffer, self)` (:69)
        const int32_t code = msj_stage1(buffer, len, structural_indexes.data(), structural_indexes.size(), &n, &verdict, flags);
        utf8_verdict = verdict;
        if (code == errors::SUCCESS || code == errors::EMPTY || code == errors::UTF8_ERROR) {
            n_structural_indexes = static_cast<uint32_t>(n);  // json_structural_indexer.mojo:160-165
            next_structural_index = 0;                        // :174
        }
        return code;
    }

    // :85-89 -- reserve(amount) + resize(amount, 0) on the list the parser keeps: memory moves only when a document
    // is larger than any before, resize zero-fills only what it adds.  Where a large list moves it is pinned
    // (msj_host_register): the indices then come down by DMA straight into it.
    void allocate(size_t amount) {
        const size_t want = amount + 3;
        if (want > structural_indexes.capacity()) {
            unregister();
            structural_indexes.reserve(want);
            if (structural_indexes.capacity() * sizeof(uint32_t) >= kRegisterFromBytes)
                _registered = msj_host_register(nullptr, structural_indexes.data(), structural_indexes.capacity() * sizeof(uint32_t)) == 0;
        }
        structural_indexes.resize(want, 0);
        _capacity = amount;
    }

    ~DomParserImplementation() { unregister(); }
    DomParserImplementation() = default;
    DomParserImplementation(const DomParserImplementation &) = delete;
    DomParserImplementation &operator=(const DomParserImplementation &) = delete;

  private:
    static constexpr size_t kRegisterFromBytes = size_t(64) << 20;
    void unregister() {
        if (_registered) (void)msj_host_unregister(nullptr, structural_indexes.data());
        _registered = false;
    }
    bool _registered = false;
    size_t _capacity = 0;
    int _max_depth = 100;  // :39
};

}  // namespace mojo_simdjson
