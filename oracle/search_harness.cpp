// search_harness.cpp -- TEST INFRASTRUCTURE.  Drives the REFERENCE's own plain-modulus recursion, plainModulusBinarySearchInternal
// (CrCNN/src/optimalParametersChooser.cpp:84-181, compiled in place by oracle/Makefile), with a table-driven verdict in place of its
// testPlainModulus (:185-226, which needs MNIST, a model and minutes of encrypted inference per candidate): the reference's definition of
// that one function is made a weak symbol in its object file (objcopy --weaken-symbol) and the definition below wins at link time -- no
// reference source is copied or edited.  Prints every candidate the reference tests, in order, and what it returns; oracle/make_golden.py
// turns that into tests/golden/search_sequences.json, the fixture the product's search (crcnn_amd/host/plain_modulus_search.cpp) is pinned to.
//   search_harness <min> <max> <pow 0|1> <first_good> <last_good>      verdict(t) = t < first_good ? MISPREDICTED : t > last_good ? OUT_OF_BUDGET : SUCCESS
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <iostream>
#include <streambuf>
#include "cnnBuilder.h"

enum exit_status_forward { SUCCESS, OUT_OF_BUDGET, MISPREDICTED };       // optimalParametersChooser.cpp:21 (not in a header there)
uint64_t plainModulusBinarySearchInternal(CnnBuilder build, uint64_t min_plain_modulus, uint64_t max_plain_modulus, int max_poly_modulus, bool pow, int num_images_to_test);

static uint64_t first_good, last_good;
exit_status_forward testPlainModulus(CnnBuilder, uint64_t plain_modulus, int, int)
{
    const exit_status_forward s = plain_modulus < first_good ? MISPREDICTED : plain_modulus > last_good ? OUT_OF_BUDGET : SUCCESS;
    fprintf(stderr, "tried %llu %s\n", (unsigned long long)plain_modulus, s == SUCCESS ? "SUCCESS" : s == OUT_OF_BUDGET ? "OUT_OF_BUDGET" : "MISPREDICTED");
    return s;
}

int main(int argc, char **argv)
{
    if (argc < 6) return 1;
    const uint64_t lo = strtoull(argv[1], 0, 0), hi = strtoull(argv[2], 0, 0); const bool pow = atoi(argv[3]) != 0;
    first_good = strtoull(argv[4], 0, 0); last_good = strtoull(argv[5], 0, 0);
    struct Null : std::streambuf { int overflow(int c) override { return c; } } null;
    std::streambuf *old = std::cout.rdbuf(&null);                        // the reference narrates every level on stdout
    CnnBuilder build("unused.h5");
    const uint64_t found = plainModulusBinarySearchInternal(build, lo, hi, 4096, pow, 2);
    std::cout.rdbuf(old);
    fprintf(stderr, "found %llu\n", (unsigned long long)found);
    return 0;
}
