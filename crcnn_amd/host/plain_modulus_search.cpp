// plain_modulus_search.cpp -- see plain_modulus_search.h.  Host logic only: every encrypted operation goes through the
// classes of crcnn_host.h (and from there through the C ABI to the gfx950 kernels).
#include "plain_modulus_search.h"
#include <algorithm>
#include <chrono>
#include <cmath>
#include <fstream>
#include <iostream>
#include <random>
#include <stdexcept>
using namespace std;

// ---- dataset helpers (CrCNN/src/utils.cpp) ------------------------------------------------------------------------------
vector<vector<float>> normalize(vector<vector<float>> dataset, float mean, float stdv)
{   // float32 arithmetic on purpose: (p / 255 - mean) / stdv as utils.cpp:13 evaluates it
    for (auto &img : dataset)
        for (float &p : img) p = (p / 255 - mean) / stdv;
    return dataset;
}
static uint32_t be32(istream &f)
{
    unsigned char b[4]; f.read((char *)b, 4);
    return ((uint32_t)b[0] << 24) | ((uint32_t)b[1] << 16) | ((uint32_t)b[2] << 8) | b[3];
}
vector<vector<float>> loadAndNormalizeMNISTestSet(string dataset_path)
{   // idx3-ubyte: magic 0x00000803, count, rows, cols, then count*rows*cols bytes (what mnist::read_dataset parses for utils.cpp:22-23)
    ifstream f(dataset_path + "/t10k-images-idx3-ubyte", ios::binary);
    if (!f) throw runtime_error("cannot open " + dataset_path + "/t10k-images-idx3-ubyte");
    if (be32(f) != 0x803) throw runtime_error("not an idx3-ubyte image file");
    const uint32_t cnt = be32(f), rows = be32(f), cols = be32(f);
    vector<vector<float>> set(cnt, vector<float>((size_t)rows * cols));
    vector<unsigned char> buf((size_t)rows * cols);
    for (uint32_t i = 0; i < cnt; i++) {
        f.read((char *)buf.data(), buf.size());
        if (!f) throw runtime_error("truncated idx3-ubyte file");
        for (size_t j = 0; j < buf.size(); j++) set[i][j] = (float)buf[j];
    }
    return normalize(set, 0.1307f, 0.3081f);
}
vector<unsigned char> loadMNISTestLabels(string dataset_path)
{
    ifstream f(dataset_path + "/t10k-labels-idx1-ubyte", ios::binary);
    if (!f) throw runtime_error("cannot open " + dataset_path + "/t10k-labels-idx1-ubyte");
    if (be32(f) != 0x801) throw runtime_error("not an idx1-ubyte label file");
    const uint32_t cnt = be32(f);
    vector<unsigned char> labels(cnt);
    f.read((char *)labels.data(), cnt);
    if (!f) throw runtime_error("truncated idx1-ubyte file");
    return labels;
}
vector<unsigned char> loadMNISTPlainModelPredictions(string file_path)
{
    ifstream file(file_path);
    vector<unsigned char> predictions;
    int label;
    while (file >> label) predictions.push_back((unsigned char)label);
    return predictions;
}

// ---- plaintext model ----------------------------------------------------------------------------------------------------
namespace {
struct Cube { int z, x, y; vector<float> v; float &at(int c, int i, int j) { return v[((size_t)c * x + i) * y + j]; } };
Cube convF(Cube &in, const vector<float> &w, const vector<float> &b, int xs, int ys, int xf, int yf, int nf)
{
    Cube o{nf, (in.x - xf) / xs + 1, (in.y - yf) / ys + 1, {}}; o.v.assign((size_t)o.z * o.x * o.y, 0.f);
    for (int f = 0; f < nf; f++)
        for (int i = 0; i < o.x; i++)
            for (int j = 0; j < o.y; j++) {
                double acc = b[f];
                for (int c = 0; c < in.z; c++)
                    for (int u = 0; u < xf; u++)
                        for (int v = 0; v < yf; v++) acc += (double)w[(((size_t)f * in.z + c) * xf + u) * yf + v] * in.at(c, i * xs + u, j * ys + v);
                o.at(f, i, j) = (float)acc;
            }
    return o;
}
Cube poolF(Cube &in, int xs, int ys, int xf, int yf, bool avg)
{
    Cube o{in.z, (in.x - xf) / xs + 1, (in.y - yf) / ys + 1, {}}; o.v.assign((size_t)o.z * o.x * o.y, 0.f);
    for (int c = 0; c < in.z; c++)
        for (int i = 0; i < o.x; i++)
            for (int j = 0; j < o.y; j++) {
                double acc = 0;
                for (int u = 0; u < xf; u++) for (int v = 0; v < yf; v++) acc += in.at(c, i * xs + u, j * ys + v);
                o.at(c, i, j) = (float)(avg ? acc / (xf * yf) : acc);
            }
    return o;
}
void bnF(Cube &t, const vector<float> &mean, const vector<float> &var)
{
    for (int c = 0; c < t.z; c++) {
        const double s = 1.0 / sqrt((double)var[c] + 0.00001);
        for (int i = 0; i < t.x * t.y; i++) t.v[(size_t)c * t.x * t.y + i] = (float)(((double)t.v[(size_t)c * t.x * t.y + i] - mean[c]) * s);
    }
}
Cube fcF(Cube &in, const vector<float> &w, const vector<float> &b, int out_dim)
{
    const size_t in_dim = in.v.size();
    Cube o{1, out_dim, 1, vector<float>(out_dim)};
    for (int r = 0; r < out_dim; r++) { double acc = b[r]; for (size_t c = 0; c < in_dim; c++) acc += (double)w[(size_t)r * in_dim + c] * in.v[c]; o.v[r] = (float)acc; }
    return o;
}
}   // namespace

vector<float> plainModelForward(CnnBuilder &build, const string &model, const vector<float> &image)
{   // the layer lists of cnnBuilder.cpp:115-169, in float
    if (image.size() != 28 * 28) throw invalid_argument("plainModelForward expects a 28x28 image");
    auto P = [&](const char *name) { return build.getPretrained(name); };
    Cube t{1, 28, 28, image};
    if (model == "PlainModelTiny") {
        t = convF(t, P("pool1_features.conv1.weight"), P("pool1_features.conv1.bias"), 1, 1, 5, 5, 32);
        t = poolF(t, 2, 2, 2, 2, true);
        t = convF(t, P("pool2_features.conv2.weight"), P("pool2_features.conv2.bias"), 1, 1, 5, 5, 64);
        t = poolF(t, 2, 2, 2, 2, true);
        t = fcF(t, P("classifier.fc3.weight"), P("classifier.fc3.bias"), 512);
        t = fcF(t, P("classifier.fc4.weight"), P("classifier.fc4.bias"), 10);
    } else if (model == "ApproxPlainModel" || model == "PlainModelWoPad") {
        const bool avg = model == "ApproxPlainModel";
        t = convF(t, P("pool1_features.conv1.weight"), P("pool1_features.conv1.bias"), 2, 2, 5, 5, 20);
        t = poolF(t, 1, 1, 2, 2, avg);
        bnF(t, P("pool1_features.norm1.running_mean"), P("pool1_features.norm1.running_var"));
        t = convF(t, P("pool2_features.conv2.weight"), P("pool2_features.conv2.bias"), 2, 2, 3, 3, 50);
        for (float &v : t.v) v = v * v;
        t = poolF(t, 1, 1, 2, 2, avg);
        bnF(t, P("pool2_features.norm2.running_mean"), P("pool2_features.norm2.running_var"));
        t = fcF(t, P("classifier.fc3.weight"), P("classifier.fc3.bias"), 500);
        t = fcF(t, P("classifier.fc4.weight"), P("classifier.fc4.bias"), 10);
    } else throw invalid_argument("unknown model " + model);
    return t.v;
}

// ---- the search -----------------------------------------------------------------------------------------------------------
// log2 through a double, truncated -- the reference's own conversion (:88-89); it rounds 2^m - 1 up to m from m = 49 on, and the
// search must visit the same moduli as the reference does
static inline uint64_t floorLog2(uint64_t v) { return (uint64_t)std::log2((double)v); }

uint64_t plainModulusBinarySearchInternal(const PlainModulusTest &test, uint64_t min_plain_modulus, uint64_t max_plain_modulus, bool pow)
{
    if (min_plain_modulus == 0 || min_plain_modulus > max_plain_modulus) throw invalid_argument("plain modulus range is empty");
    // positions are exponents (pow) or the moduli themselves; the reference converts with log2() truncated to an integer (:87-90)
    const uint64_t lo = pow ? floorLog2(min_plain_modulus) : min_plain_modulus;
    const uint64_t hi = pow ? floorLog2(max_plain_modulus) : max_plain_modulus;
    auto value = [&](uint64_t pos) { return pow ? (uint64_t)1 << pos : pos; };

    if (hi - lo <= 1) {                                      // base of the recursion (:91-128): only the two ends are left
        const exit_status_forward s = test(value(lo));
        if (s == SUCCESS) return value(lo);
        if (s == OUT_OF_BUDGET) return 0;                    // the larger end can only have less budget
        if (hi - lo == 1) return test(value(hi)) == SUCCESS ? value(hi) : 0;
        return 0;
    }
    const uint64_t t = value(lo + (hi - lo) / 2);
    const exit_status_forward s = test(t);
    if (s == SUCCESS || s == OUT_OF_BUDGET) {                // smaller moduli have more budget: look left (:145-162)
        const uint64_t below = plainModulusBinarySearchInternal(test, value(lo), t - 1, pow);
        if (below > 0) return below;
        return s == SUCCESS ? t : 0;
    }
    if (t >= value(hi)) return 0;                            // mispredicted (:165-175): larger moduli give the encoding more room
    return plainModulusBinarySearchInternal(test, t + 1, value(hi), pow);
}

uint64_t plainModulusBinarySearch(const PlainModulusTest &test, uint64_t min_plain_modulus, uint64_t max_plain_modulus, uint64_t min_prime_coeff_mod)
{
    const uint64_t found = plainModulusBinarySearchInternal(test, min_plain_modulus, max_plain_modulus, true);
    if (found > 0 && found >= min_prime_coeff_mod) {
        // t >= some q_i switches the fast plain lift off (:44-58); try every integer of the top binade below the smallest prime
        const uint64_t fast = plainModulusBinarySearchInternal(test, (uint64_t)1 << floorLog2(min_prime_coeff_mod), min_prime_coeff_mod - 1, false);
        if (fast > 0) return fast;
    }
    return found;
}

uint64_t minSmallModulusinCoeffModulus(int max_poly_modulus)
{
    uint64_t q[16];
    const int k = crc_default_coeff_modulus_128(max_poly_modulus, q, 16);
    if (k <= 0) throw invalid_argument("no default coeff_modulus for this poly_modulus");
    return *min_element(q, q + k);
}

void PlainModulusSearch::predictWithPlainModel(const string &path_to_model)
{
    CnnBuilder build(path_to_model);
    predicted_labels.resize(test_set.size());
    for (size_t i = 0; i < test_set.size(); i++) {
        const vector<float> logits = plainModelForward(build, model, test_set[i]);
        predicted_labels[i] = (unsigned char)(max_element(logits.begin(), logits.end()) - logits.begin());
    }
}

exit_status_forward PlainModulusSearch::testPlainModulus(CnnBuilder &build, uint64_t plain_modulus, int num_images_to_test)
{
    if (test_set.empty() || predicted_labels.size() < test_set.size()) throw logic_error("PlainModulusSearch needs a test set and its plaintext predictions");
    const auto t0 = chrono::high_resolution_clock::now();
    // same image picks as the reference for a given seed (:185-186, :199); its distribution's upper end is one past the last image
    default_random_engine generator(seed);
    uniform_int_distribution<int> distribution(0, (int)predicted_labels.size());
    exit_status_forward ret_value = SUCCESS;

    if (coeff_modulus.empty()) setParameters(max_poly_modulus, plain_modulus);
    else setParameters(max_poly_modulus, coeff_modulus, plain_modulus, 0);
    Network net = build.buildNetworkByName(model);
    net.max_num_of_reencryptions = max_num_of_reencryptions;             // budget-checking forward (network.cpp:52-96)

    vector<int> picks; vector<ciphertext3D> enc;
    for (int i = 0; i < num_images_to_test; i++) {
        const int img_test = min(distribution(generator), (int)test_set.size() - 1);
        picks.push_back(img_test);
        enc.push_back(encryptImage(test_set[img_test], 1, 28, 28));
    }
    try {
        // all picked images go through the network as one batch; the reference runs them one after the other and stops at the
        // first failure, which yields the same verdict (the budget does not depend on the image)
        const ciphertext3D out = net.forward(stackImages(enc));
        const vector<floatCube> logits = decryptImages(out);
        for (size_t i = 0; i < logits.size() && ret_value == SUCCESS; i++) {
            int predicted = 0;
            for (int j = 1; j < (int)logits[i][0].size(); j++) if (logits[i][0][j][0] > logits[i][0][predicted][0]) predicted = j;
            if (predicted != predicted_labels[picks[i]]) ret_value = MISPREDICTED;
        }
    } catch (const OutOfBudgetException &) {
        ret_value = OUT_OF_BUDGET;
    }
    delParameters();
    tried.emplace_back(plain_modulus, ret_value);
    test_seconds.push_back(chrono::duration<double>(chrono::high_resolution_clock::now() - t0).count());
    return ret_value;
}

uint64_t PlainModulusSearch::run(int num_images_to_test, uint64_t min_plain_modulus, uint64_t max_plain_modulus, const string &path_to_model)
{
    CnnBuilder build(path_to_model);
    uint64_t min_q = 0;
    if (coeff_modulus.empty()) min_q = minSmallModulusinCoeffModulus(max_poly_modulus);
    else min_q = *min_element(coeff_modulus.begin(), coeff_modulus.end());
    return plainModulusBinarySearch([&](uint64_t t) { return testPlainModulus(build, t, num_images_to_test); }, min_plain_modulus, max_plain_modulus, min_q);
}

uint64_t plainModulusBinarySearch(int num_images_to_test, uint64_t min_plain_modulus, uint64_t max_plain_modulus, string path_to_model)
{
    PlainModulusSearch s;
    s.test_set = loadAndNormalizeMNISTestSet("../PlainModel/MNISTdata/raw");
    s.predicted_labels = loadMNISTPlainModelPredictions("../PlainModel/predictionsApproxPlainModel.csv");
    return s.run(num_images_to_test, min_plain_modulus, max_plain_modulus, path_to_model);
}
