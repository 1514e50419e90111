// C++ twin of the reference's own stage-1 test, tests/test_stage_1.mojo, on top of the C++ mirror of its
// facade (include/dom_parser_implementation.hpp): same helper names, same checks, same fixtures
// (tests/golden/jsons_for_test = a copy of the reference's tests/jsons_for_test data files).
//   usage: test_stage_1 <jsons_for_test directory>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <filesystem>
#include <fstream>
#include <stdexcept>
#include <string>
#include <vector>

#include "dom_parser_implementation.hpp"

using mojo_simdjson::DomParserImplementation;
namespace fs = std::filesystem;

static void assert_true(bool v, const char *what) {
    if (!v) throw std::runtime_error(std::string("assertion failed: ") + what);
}

// tests/test_stage_1.mojo:22-24
static void assert_strictly_increasing(const std::vector<uint32_t> &array, uint32_t length) {
    for (uint32_t i = 1; i < length; i++) assert_true(array[i - 1] < array[i], "strictly increasing");
}

// :27-40
static void assert_tagging_is_correct(const std::string &json_input, const std::string &expected) {
    const std::string structural = "{}[]:,tfn-0123456789\"";
    for (size_t i = 0; i < std::min(json_input.size(), expected.size()); i++) {
        if (expected[i] != '1') continue;
        if (structural.find(json_input[i]) == std::string::npos)
            throw std::runtime_error(std::string("Wrong tagging of characters, ") + json_input[i] +
                                     " is not a structural character");
    }
}

// :43-82
static void verify_expected_structural_characters(const DomParserImplementation &parser, const std::string &expected,
                                                  const std::string &json_input) {
    assert_tagging_is_correct(json_input, expected);
    assert_strictly_increasing(parser.structural_indexes, parser.n_structural_indexes);
    std::string detected(expected.size(), ' ');
    for (uint32_t i = 0; i < parser.n_structural_indexes; i++) detected[parser.structural_indexes[i]] = '1';
    if (detected != expected) {
        std::printf("Error in the detected structural characters\n%s\n%s expected\n%s detected\n", json_input.c_str(),
                    expected.c_str(), detected.c_str());
        throw std::runtime_error("Detected and expected structural characters do not match");
    }
    // 3 are leftover
    const uint32_t n = parser.n_structural_indexes;
    assert_true(parser.structural_indexes[n] == json_input.size(), "trailer[0] == len");
    assert_true(parser.structural_indexes[n + 1] == json_input.size(), "trailer[1] == len");
    assert_true(parser.structural_indexes[n + 2] == 0, "trailer[2] == 0");
}

// :85-96
static void check_stage1(const fs::path &json_file) {
    std::ifstream in(json_file, std::ios::binary);
    std::string json_input, expected;
    std::getline(in, json_input);
    std::getline(in, expected);
    DomParserImplementation parser;
    const int error_code = parser.stage1(json_input);
    if (error_code != 0) throw std::runtime_error("unexpected error code " + std::to_string(error_code));
    verify_expected_structural_characters(parser, expected, json_input);
}

static bool raises(const fs::path &file, const std::string &contains) {
    try {
        check_stage1(file);
    } catch (const std::runtime_error &e) {
        return std::string(e.what()).find(contains) != std::string::npos;
    }
    return false;
}

int main(int argc, char **argv) {
    if (argc < 2) return 2;
    const fs::path dir = argv[1];
    int failures = 0;
    // test_wrong_tagging (:99-102), test_detect_incorrect_result (:105-110): the harness itself
    if (!raises(dir / "wrong_tagging.json", "l is not a structural character")) {
        std::printf("FAIL test_wrong_tagging\n");
        failures++;
    }
    if (!raises(dir / "detect_incorrect_result.json", "Detected and expected structural characters do not match")) {
        std::printf("FAIL test_detect_incorrect_result\n");
        failures++;
    }
    // test_simple_json (:113-122)
    int files = 0;
    for (const auto &entry : fs::directory_iterator(dir / "valid")) {
        if (!entry.is_regular_file()) continue;
        files++;
        try {
            check_stage1(entry.path());
        } catch (const std::exception &e) {
            std::printf("FAIL %s: %s\n", entry.path().filename().c_str(), e.what());
            failures++;
        }
    }
    if (files <= 5) {
        std::printf("FAIL: only %d fixture files\n", files);
        failures++;
    }
    std::printf("%s: %d fixture files, %d failures\n", failures ? "FAILED" : "test_stage_1 ok", files, failures);
    return failures ? 1 : 0;
}
