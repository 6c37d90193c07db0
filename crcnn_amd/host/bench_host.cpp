// bench_host.cpp -- the C++ host classes (crcnn_host.h: the drop-in for the reference's Layer / Network / CnnBuilder) as a MEASURED path: what bench.py times through
// its Python twin (netrun.py), timed here through Network::forward.  The reference's timed driver is CrCNN/src/mainparams.cpp:64-116 (one image at a time, chrono
// around every layer -> the T_LAYER_i columns, mainparams.cpp:81); this is the same loop over chunks of encrypted images.
//   bench_host <model> <model.h5> <n> <k> <t> <inputs.u64> <distinct> <batch> <chunk> <steps> <out0.u64> [group]
// group > 1: Network::forward gets chunk * group images with head_chunk = chunk (two-level chunking: the dense layers once per group)
// inputs.u64: `distinct` encrypted images ([distinct][784][2][k][n] u64, coefficient form: bench.py writes the very ciphertexts it runs itself), tiled to the chunk.
// Prints one JSON line: images/s over `steps` passes of `batch` images, T_LAYER_i in ms per image (wall clock around Layer::forward + stream sync, as the reference
// measures), the kernel each conv / dense layer ran on; writes the 10 output ciphertexts of image 0 to <out0.u64> (bench.py compares their SHA-256 with the golden).
#include "crcnn_host.h"
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <iostream>
using namespace std;

int main(int argc, char **argv)
{
    if (argc < 12) { fprintf(stderr, "usage: %s <model> <h5> <n> <k> <t> <inputs.u64> <distinct> <batch> <chunk> <steps> <out0.u64>\n", argv[0]); return 1; }
    const string model = argv[1], h5 = argv[2], inputs = argv[6], out0 = argv[11];
    const int n = atoi(argv[3]), k = atoi(argv[4]); const uint64_t t = strtoull(argv[5], 0, 0);
    const int distinct = atoi(argv[7]), batch = atoi(argv[8]), head = atoi(argv[9]), steps = atoi(argv[10]), group = argc > 12 ? max(1, atoi(argv[12])) : 1;
    const int chunk = head * group;
    try {
        uint64_t q[16];
        const int kd = crc_default_coeff_modulus_128(n, q, 16);
        if (kd < k) throw invalid_argument("coeff_modulus_128(n) has fewer primes than asked for");
        const auto t_setup = chrono::high_resolution_clock::now();
        setDeterministicSeed(2024);                              // (keys are not used on the timed path; the inputs come encrypted)
        setParameters(n, vector<uint64_t>(q, q + k), t, 0);
        CnnBuilder build(h5);
        Network net = build.buildNetworkByName(model);
        net.fuse();
        if (group > 1) net.head_chunk = head;
        // the chunk: `distinct` images tiled
        const size_t ctw = (size_t)2 * k * n, imgw = 784 * ctw;
        vector<uint64_t> h((size_t)distinct * imgw);
        { ifstream f(inputs, ios::binary); if (!f) throw runtime_error("cannot open " + inputs); f.read((char *)h.data(), (streamsize)(h.size() * 8)); if (!f) throw runtime_error("short read: " + inputs); }
        vector<ciphertext3D> one;
        for (int d = 0; d < distinct; d++) one.push_back(ciphertext3D::fromHost(h.data() + (size_t)d * imgw, 1, 1, 28, 28));
        vector<ciphertext3D> tiled;
        for (int b = 0; b < chunk; b++) tiled.push_back(one[b % distinct]);
        const ciphertext3D x = stackImages(tiled);
        tiled.clear(); one.clear(); h.clear(); h.shrink_to_fit();
        // untimed first pass: operand forms, module load; image 0's outputs
        {
            ciphertext3D y = net.forward(x);
            vector<uint64_t> yh = y.toHost();
            ofstream o(out0, ios::binary); o.write((const char *)yh.data(), (streamsize)(10 * ctw * 8));
        }
        const double setup_s = chrono::duration<double>(chrono::high_resolution_clock::now() - t_setup).count();
        const int L = net.getNumLayers();
        vector<double> tl(L, 0.0);
        const int chunks = batch / chunk;
        const auto t0 = chrono::high_resolution_clock::now();
        for (int s = 0; s < steps; s++)
            for (int c = 0; c < chunks; c++) {
                ciphertext3D y = net.forward(x);
                for (int i = 0; i < L; i++) tl[i] += net.last_layer_ms[i];
            }
        if (crc_stream_sync(context, nullptr) < 0) throw runtime_error("crc_stream_sync");
        const double dt = chrono::duration<double>(chrono::high_resolution_clock::now() - t0).count();
        const double images = (double)steps * chunks * chunk;
        printf("{\"host\": \"C++ classes of crcnn_amd/host (Network::forward)\", \"model\": \"%s\", \"n\": %d, \"k\": %d, \"batch\": %d, \"chunk\": %d, \"group\": %d, \"steps\": %d, \"images_per_s\": %.4f, "
               "\"ms_per_image\": %.4f, \"setup_s\": %.1f, \"T_LAYER_ms_per_image\": [", model.c_str(), n, k, chunks * chunk, head, group, steps, images / dt, dt / images * 1e3, setup_s);
        for (int i = 0; i < L; i++) printf("%s%.4f", i ? ", " : "", tl[i] / images);
        printf("], \"layers\": [");
        for (int i = 0; i < L; i++) printf("%s\"%s\"", i ? ", " : "", net.getLayer(i)->getName().c_str());
        printf("]}\n");
        delParameters();
        return 0;
    } catch (const exception &e) {
        fprintf(stderr, "bench_host: %s\n", e.what());
        return 2;
    }
}
