// bench_host.cpp -- the MEASURED path of bench.py: the C++ host classes (crcnn_host.h: the drop-in for the reference's Layer / Network / CnnBuilder) over the C ABI.
// The reference's timed driver is CrCNN/src/mainparams.cpp:64-116 (one image at a time, chrono around every layer -> the T_LAYER_i columns, mainparams.cpp:81);
// this is the same loop over chunks of encrypted images: CnnBuilder builds the network from the HDF5 model, Network::fuse() folds it, Network::forward runs it.
//
//   bench_host model=<name> h5=<model.h5> n=<n> k=<k> t=<t> [q=<p0>,<p1>,..] inputs=<file> distinct=<D> batch=<B> chunk=<C> group=<G> steps=<K> warmup=<W> outputs=<file> [fuse=1]
//
// inputs   D encrypted images ([D][784][2][k][n] u64, coefficient form: bench.py's client side writes them), tiled to one launch of C * G images that every chunk
//          of the batch re-reads (the same bytes per image as a resident batch; bench.py states it under "data")
// group    > 1: two-level chunking -- Network::forward gets C * G images with head_chunk = C: the layers in front of the first dense layer per chunk, the dense
//          layers once per group
// timing   W untimed passes over the batch, then K timed ones bracketed by stream synchronisations; per-layer times from HIP events on the launch stream
//          (Network::time_with_events: no synchronisation between layers)
// outputs  the 10 output ciphertexts of the first D images of an untimed launch ([D][10][2][k][n] u64): bench.py hashes them against the reference's goldens and
//          decrypts them
// Prints ONE JSON line on stdout.
#include "crcnn_host.h"
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iostream>
#include <map>
#include <unistd.h>
using namespace std;

static string jstr(const string &s) { string o = "\""; for (char c : s) { if (c == '"' || c == '\\') o += '\\'; o += c; } return o + "\""; }

int main(int argc, char **argv)
{
    map<string, string> a;
    for (int i = 1; i < argc; i++) { const char *eq = strchr(argv[i], '='); if (!eq) { fprintf(stderr, "bench_host: argument without '=': %s\n", argv[i]); return 1; } a[string(argv[i], (size_t)(eq - argv[i]))] = eq + 1; }
    auto need = [&](const char *k) -> string { if (!a.count(k)) { fprintf(stderr, "bench_host: missing %s=\n", k); exit(1); } return a[k]; };
    auto geti = [&](const char *k, long long d) { return a.count(k) ? atoll(a[k].c_str()) : d; };
    const string model = need("model"), h5 = need("h5"), inputs = need("inputs"), outputs = need("outputs");
    const int n = (int)geti("n", 0), k = (int)geti("k", 0); const uint64_t t = strtoull(need("t").c_str(), 0, 0);
    const int distinct = (int)geti("distinct", 1), batch = (int)geti("batch", 1), head = (int)geti("chunk", 1), group = max(1, (int)geti("group", 1));
    const int steps = (int)geti("steps", 1), warmup = (int)geti("warmup", 0), fuse = (int)geti("fuse", 1);
    const int launch = head * group;                          // images per Network::forward
    try {
        if (n < 1 || k < 1 || launch < 1 || batch < launch || distinct < 1 || distinct > launch || steps < 1) throw invalid_argument("bad sizes (need 1 <= distinct <= chunk * group <= batch)");
        uint64_t q[16];
        if (a.count("q")) {                                      // explicit coefficient modulus (q=<prime>,<prime>,...); default: the first k primes of coeff_modulus_128(n)
            int cnt = 0; const char *p = a["q"].c_str();
            while (*p && cnt < 16) { char *e; q[cnt++] = strtoull(p, &e, 0); p = *e == ',' ? e + 1 : e; if (e == p && *e) break; }
            if (cnt != k) throw invalid_argument("q= must list k primes");
        } else {
            const int kd = crc_default_coeff_modulus_128(n, q, 16);
            if (kd < k) throw invalid_argument("coeff_modulus_128(n) has fewer primes than asked for");
        }
        const auto t_setup = chrono::high_resolution_clock::now();
        setDeterministicSeed((uint64_t)geti("key_seed", 2024));   // the seeded client side of bench.py: the same evaluation keys, hence the same ciphertexts behind Square
        setParameters(n, vector<uint64_t>(q, q + k), t, (int)geti("device", 0));
        size_t free0 = 0, total = 0;
        crc_mem_info(context, &free0, &total);
        CnnBuilder build(h5);
        Network net = build.buildNetworkByName(model);
        if (fuse) net.fuse();
        if (group > 1) net.head_chunk = head;
        net.time_with_events = true;
        // one launch's images: the D distinct ones tiled
        const size_t ctw = (size_t)2 * k * n, imgw = 784 * ctw;
        ciphertext3D x;
        {
            vector<uint64_t> h((size_t)distinct * imgw);
            ifstream f(inputs, ios::binary); if (!f) throw runtime_error("cannot open " + inputs);
            f.read((char *)h.data(), (streamsize)(h.size() * 8)); if (!f) throw runtime_error("short read: " + inputs);
            vector<ciphertext3D> one, tiled;
            for (int d = 0; d < distinct; d++) one.push_back(ciphertext3D::fromHost(h.data() + (size_t)d * imgw, 1, 1, 28, 28));
            for (int b = 0; b < launch; b++) tiled.push_back(one[b % distinct]);
            x = stackImages(tiled);
        }
        // untimed first launch: operand forms, module load; the distinct images' outputs
        {
            ciphertext3D y = net.forward(x);
            vector<uint64_t> yh = y.toHost();
            ofstream o(outputs, ios::binary); o.write((const char *)yh.data(), (streamsize)((size_t)distinct * 10 * ctw * 8));
            if (!o) throw runtime_error("cannot write " + outputs);
        }
        const double setup_s = chrono::duration<double>(chrono::high_resolution_clock::now() - t_setup).count();
        size_t free1 = 0; crc_mem_info(context, &free1, &total);
        const Network::HbmPlan plan = net.hbmPlan();
        const int L = net.getNumLayers();
        const int launches = batch / launch;
        for (int s = 0; s < warmup; s++) for (int c = 0; c < launches; c++) net.forward(x);
        if (crc_stream_sync(context, nullptr) < 0) throw runtime_error("crc_stream_sync");
        vector<double> tl(L, 0.0); vector<long long> calls(L, 0);
        const auto t0 = chrono::high_resolution_clock::now();
        for (int s = 0; s < steps; s++)
            for (int c = 0; c < launches; c++) {
                ciphertext3D y = net.forward(x);
                for (int i = 0; i < L; i++) { tl[i] += net.last_layer_ms[i]; calls[i] += net.last_layer_launches[i]; }
            }
        if (crc_stream_sync(context, nullptr) < 0) throw runtime_error("crc_stream_sync");
        const double dt = chrono::duration<double>(chrono::high_resolution_clock::now() - t0).count();
        const double images = (double)steps * launches * launch;
        printf("{\"host\": \"C++ host classes (crcnn_amd/host: CnnBuilder / Network::fuse / Network::forward) over the C ABI\", \"model\": %s, \"n\": %d, \"k\": %d, \"batch\": %d, \"chunk\": %d, "
               "\"group\": %d, \"steps\": %d, \"warmup\": %d, \"images_per_s\": %.4f, \"elapsed_s\": %.6f, \"ms_per_step\": %.3f, \"ms_per_image\": %.4f, \"setup_s\": %.1f, ",
               jstr(model).c_str(), n, k, launches * launch, head, group, steps, warmup, images / dt, dt, dt / steps * 1e3, dt / images * 1e3, setup_s);
        printf("\"layers\": [");
        for (int i = 0; i < L; i++) printf("%s%s", i ? ", " : "", jstr(net.getLayer(i)->getName()).c_str());
        printf("], \"kernel_per_layer\": [");
        for (int i = 0; i < L; i++) printf("%s%s", i ? ", " : "", jstr(net.getLayer(i)->kernelName()).c_str());
        printf("], \"T_LAYER_ms_per_image\": [");
        for (int i = 0; i < L; i++) printf("%s%.4f", i ? ", " : "", tl[i] / images);
        printf("], \"layer_launch_ms\": [");
        for (int i = 0; i < L; i++) printf("%s%.4f", i ? ", " : "", calls[i] ? tl[i] / calls[i] : 0.0);
        printf("], \"layer_launches\": [");
        for (int i = 0; i < L; i++) printf("%s%lld", i ? ", " : "", calls[i]);
        printf("], \"layer_timing\": \"HIP events on the launch stream around every Layer::forward inside the timed region\", ");
        printf("\"hbm\": {\"total_bytes\": %zu, \"free_before_build\": %zu, \"free_after_first_launch\": %zu, \"parameters\": %zu, \"activation_slots\": %zu, \"work_buffer\": %zu, "
               "\"evaluation_keys\": %zu, \"input_launch\": %zu}}\n", total, free0, free1, plan.parameters, plan.activations, plan.work, plan.keys, (size_t)launch * imgw * 8);
        delParameters();
        return 0;
    } catch (const exception &e) {
        fprintf(stderr, "bench_host: %s\n", e.what());
        fflush(stderr);
        _exit(2);                                               // (no unwinding of device state after a failed launch or allocation: the message above is the report)
    }
}
