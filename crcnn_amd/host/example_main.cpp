// example_main.cpp -- a CrCNN-style driver on the MI355X engine: what CrCNN/src/mainparams.cpp:64-116 does (set parameters, build the
// network from the HDF5 model, then per image: encrypt, Network::forward, decrypt, compare the arg-max with the plaintext model's),
// written against crcnn_host.h / plain_modulus_search.h only.  It prints the reference's CSV columns (mainparams.cpp:81):
//   OUTPUT: <index>,<T_LAYER_0 ms>,...,<T_REENC ms>,<T_LAYER_r ms>,...,<prediction>,<Success|Mispredicted|Out of Budget>
// T_REENC (the client-side refresh of network.cpp:29-37) stands where the reference prints it -- in front of the layer it precedes -- when a refresh layer is set;
// with the budget-checking forward (max re-encryptions) it is the total of the refreshes that ran and stands after the layer columns.
// usage: example_main <model name> <model.h5> <images.f32 (N x 784 normalised float32)> <poly_modulus> <plain_modulus> <num images> [fuse 0|1] [layer_before_reenc] [max_reencryptions]
#include "crcnn_host.h"
#include "plain_modulus_search.h"
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <iostream>
using namespace std;

int main(int argc, char **argv)
{
    if (argc < 7) { fprintf(stderr, "usage: %s <model> <h5> <images.f32> <n> <t> <num> [fuse]\n", argv[0]); return 1; }
    const string model = argv[1], h5 = argv[2], images = argv[3];
    const int poly_modulus = atoi(argv[4]); const uint64_t plain_modulus = strtoull(argv[5], 0, 0);
    const int num_images_to_test = atoi(argv[6]); const bool fuse = argc > 7 && atoi(argv[7]);
    try {
        ifstream f(images, ios::binary);
        if (!f) throw runtime_error("cannot open " + images);
        f.seekg(0, ios::end); const size_t cnt = (size_t)f.tellg() / (784 * 4); f.seekg(0);
        vector<vector<float>> test_set(cnt, vector<float>(784));
        for (auto &im : test_set) f.read((char *)im.data(), 784 * 4);

        setParameters(poly_modulus, plain_modulus);
        CnnBuilder build(h5);
        Network net = build.buildNetworkByName(model);
        if (fuse) net.fuse();
        if (argc > 8) net.layer_before_reenc = atoi(argv[8]);
        if (argc > 9) net.max_num_of_reencryptions = atoi(argv[9]);
        const int reenc_at = net.max_num_of_reencryptions >= 0 ? net.getNumLayers() : net.layer_before_reenc;      // column position of T_REENC (-1: no refresh, no column)
        cout << "INDEX_IMG";
        for (int i = 0; i < net.getNumLayers(); i++) { if (i == reenc_at) cout << ",T_REENC"; cout << ",T_LAYER_" << i; }
        if (reenc_at == net.getNumLayers()) cout << ",T_REENC";
        cout << ",PREDICTION" << endl;
        int ok = 0;
        for (int i = 0; i < num_images_to_test && i < (int)cnt; i++) {
            const vector<float> logits_plain = plainModelForward(build, model, test_set[i]);
            const int expected = (int)(max_element(logits_plain.begin(), logits_plain.end()) - logits_plain.begin());
            cout << "OUTPUT: " << i << ",";
            ciphertext3D encrypted_image = encryptImage(test_set[i], 1, 28, 28);
            exit_status_forward ret_value = SUCCESS;
            int predicted = -1;
            try {
                encrypted_image = net.forward(encrypted_image);
                for (int l = 0; l < (int)net.last_layer_ms.size(); l++) { if (l == reenc_at) cout << net.last_reenc_ms << ","; cout << net.last_layer_ms[l] << ","; }
                if (reenc_at == (int)net.last_layer_ms.size()) cout << net.last_reenc_ms << ",";
                if (noiseBudget(encrypted_image) <= 0) ret_value = OUT_OF_BUDGET;
                floatCube image = decryptImage(encrypted_image);
                predicted = 0;
                for (int j = 1; j < (int)image[0].size(); j++) if (image[0][j][0] > image[0][predicted][0]) predicted = j;
                if (ret_value == SUCCESS && predicted != expected) ret_value = MISPREDICTED;
            } catch (const OutOfBudgetException &e) {
                cout << "Maximum layer computed is " << e.last_layer_computed << " exit due to OUT_OF_BUDGET" << endl;
                ret_value = OUT_OF_BUDGET;
            }
            cout << predicted << "," << (ret_value == SUCCESS ? "Success" : (ret_value == OUT_OF_BUDGET ? "Out of Budget" : "Mispredicted")) << endl;
            ok += ret_value == SUCCESS;
        }
        delParameters();
        cout << "SUMMARY: " << ok << " of " << min<int>(num_images_to_test, (int)cnt) << " images predicted like the plaintext model" << endl;
        return 0;
    } catch (const exception &e) {
        fprintf(stderr, "error: %s\n", e.what());
        return 2;
    }
}
