// plain_modulus_search.h -- the reference's parameter-selection tool on top of the GPU engine.
//
// Mirrors CrCNN/src/optimalParametersChooser.{h,cpp} (binary search for the smallest plain modulus t at which a network still
// predicts like its plaintext model without exhausting the noise budget) and the dataset helpers of CrCNN/src/utils.{h,cpp}.
// The search itself is a pure function of a predicate t -> {SUCCESS, OUT_OF_BUDGET, MISPREDICTED}, so its control flow is
// testable without a GPU; PlainModulusSearch binds that predicate to setParameters / buildNetwork / encryptImage /
// Network::forward / decryptImage exactly as testPlainModulus (optimalParametersChooser.cpp:183-226) does.
#pragma once
#include "crcnn_host.h"
#include <functional>
#include <string>
#include <utility>
#include <vector>

// ---- CrCNN/src/utils.h ----------------------------------------------------------------------------------------------------
std::vector<std::vector<float>> normalize(std::vector<std::vector<float>> dataset, float mean, float stdv);          // utils.cpp:9-17
std::vector<std::vector<float>> loadAndNormalizeMNISTestSet(std::string dataset_path);      // <path>/t10k-images-idx3-ubyte, utils.cpp:20-30
std::vector<unsigned char> loadMNISTestLabels(std::string dataset_path);                    // <path>/t10k-labels-idx1-ubyte, utils.cpp:32-39
std::vector<unsigned char> loadMNISTPlainModelPredictions(std::string file_path);           // one label per line, utils.cpp:41-54

// float32 forward of the plaintext network behind a model file (what PlainModel/*.py computes): the source of the "plain model
// predictions" when no predictions*.csv exists for the images at hand (synthetic inputs).  Returns the 10 logits.
std::vector<float> plainModelForward(CnnBuilder &build, const std::string &model, const std::vector<float> &image);

// ---- CrCNN/src/optimalParametersChooser.cpp --------------------------------------------------------------------------------
enum exit_status_forward { SUCCESS, OUT_OF_BUDGET, MISPREDICTED };                           // optimalParametersChooser.cpp:21
typedef std::function<exit_status_forward(uint64_t /*plain_modulus*/)> PlainModulusTest;

// One level of the search (:73-180).  pow = true walks the exponents of powers of two between min and max, pow = false every
// integer.  SUCCESS and OUT_OF_BUDGET both send the search to smaller moduli (more budget), MISPREDICTED to larger ones.
// Returns the modulus found or 0.
uint64_t plainModulusBinarySearchInternal(const PlainModulusTest &test, uint64_t min_plain_modulus, uint64_t max_plain_modulus, bool pow);
// The two-phase driver (:30-60): powers of two first; if the result is not below every coefficient prime (fast plain lift off,
// SEAL context.cpp:156-165) a second search over all integers of [2^floor(log2 min_q), min_q - 1].
uint64_t plainModulusBinarySearch(const PlainModulusTest &test, uint64_t min_plain_modulus, uint64_t max_plain_modulus, uint64_t min_prime_coeff_mod);
uint64_t minSmallModulusinCoeffModulus(int max_poly_modulus);                                // :62-70

struct PlainModulusSearch {
    // what the reference keeps in file-level globals (:16-19) and hard-codes inside the functions (:32, :41)
    std::string model = "ApproxPlainModel";                  // topology behind path_to_model (cnnBuilder.cpp:108-179 selects it by source edit)
    int max_poly_modulus = 4096;
    std::vector<uint64_t> coeff_modulus;                     // empty: coeff_modulus_128(max_poly_modulus)
    std::vector<std::vector<float>> test_set;                // normalised images
    std::vector<unsigned char> predicted_labels;             // plaintext-model prediction per image
    unsigned seed = 0;
    int max_num_of_reencryptions = 0;                        // refreshes Network::forward may spend before it gives up (network.cpp:57)
    std::vector<std::pair<uint64_t, exit_status_forward>> tried;     // every modulus tested, in order
    std::vector<double> test_seconds;                        // wall time of each test

    // labels from the plaintext model itself (for image sets without a predictions file)
    void predictWithPlainModel(const std::string &path_to_model);
    exit_status_forward testPlainModulus(CnnBuilder &build, uint64_t plain_modulus, int num_images_to_test);         // :183-226
    uint64_t run(int num_images_to_test, uint64_t min_plain_modulus, uint64_t max_plain_modulus, const std::string &path_to_model);
};

// the reference's entry point (:30): MNIST test images and predictionsApproxPlainModel.csv from ../PlainModel, n = 4096
uint64_t plainModulusBinarySearch(int num_images_to_test, uint64_t min_plain_modulus, uint64_t max_plain_modulus, std::string path_to_model);
