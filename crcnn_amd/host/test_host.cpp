// test_host.cpp -- driver for the C++ host classes, run by tests/test_gpu_host_cpp.py on the GPU box.
//   test_host net <model> <h5> <dir> <resident 0|1> <batch> [fuse 0|1]      (fuse: Network::fuse() before the resident forward)
//     <dir>/params.u64 (n,k,t,q...), evk.u64, net_in.u64 ([1][1][28][28][2][k][n]) -> writes layer_<i>.u64 (layerwise mode) and out.u64
//   test_host api <h5> <dir>     exercises save/load of the encoded model, client-side encrypt/decrypt, and error behaviour
//   test_host files <dir>        CrCNN's own files: loads the encoded-model stream and the cipher_image file the REFERENCE wrote (<dir>/ref_encoded_layers.bin,
//     ref_cipher_image.bin; cnnBuilder.cpp:181-196, globals.cpp:174-205), runs conv -> bn -> dense on them (out_from_ref_files.u64), then writes the same two
//     files itself (our_encoded_layers.bin, our_cipher_image.bin) and runs those (out_from_our_files.u64)
//   test_host searchlogic <min> <max> <first_good> <last_good> <min_q>
//     the plain-modulus search on a synthetic predicate (t < first_good: MISPREDICTED, t > last_good: OUT_OF_BUDGET); no GPU work.
//     prints "found <t>" and one "tried <t> <status>" line per test
//   test_host search <model> <h5> <images.f32> <n> <min> <max> <num_images> <seed> [q0 q1 ...]   (no q: coeff_modulus_128(n))
//     the real search: images.f32 = N x 784 normalised float32 pixels, labels from the float model; prints found / tried lines
#include "crcnn_host.h"
#include "plain_modulus_search.h"
#include <cmath>
#include <cstdio>
#include <sstream>
#include <unistd.h>
#include <cstring>
#include <fstream>
#include <iostream>
using namespace std;
typedef uint64_t u64;

static vector<u64> rd(const string &p)
{
    ifstream f(p, ios::binary); if (!f) { fprintf(stderr, "missing %s\n", p.c_str()); exit(2); }
    f.seekg(0, ios::end); size_t sz = f.tellg(); f.seekg(0); vector<u64> v(sz / 8); f.read((char *)v.data(), sz); return v;
}
static void wr(const string &p, const vector<u64> &v) { ofstream f(p, ios::binary); f.write((const char *)v.data(), v.size() * 8); }

static void setup(const string &dir)
{
    auto p = rd(dir + "/params.u64");
    int n = (int)p[0], k = (int)p[1]; u64 t = p[2];
    setParameters(n, vector<u64>(p.begin() + 3, p.begin() + 3 + k), t, 0);
}

static int do_net(int argc, char **argv)
{
    if (argc < 7) return 1;
    string model = argv[2], h5 = argv[3], dir = argv[4]; bool resident = atoi(argv[5]); int batch = atoi(argv[6]);
    setup(dir);
    ifstream evf(dir + "/evk.u64", ios::binary);
    if (evf) {            // use the caller's evaluation keys (ev_keys16 is a public global in the reference as well, globals.h:26)
        auto evk = rd(dir + "/evk.u64");
        ev_keys16 = make_shared<DeviceBuffer>(evk.size() * 8);
        crc_memcpy_h2d(context, ev_keys16->ptr, evk.data(), evk.size() * 8, nullptr); crc_stream_sync(context, nullptr);
    }
    CnnBuilder builder(h5);
    Network net = builder.buildNetworkByName(model);
    net.ntt_resident = resident;
    if (argc > 7 && atoi(argv[7])) { const int removed = net.fuse(); fprintf(stderr, "fused: %d layers removed, %d left\n", removed, net.getNumLayers()); }
    // two-level chunking: the layers in front of the first dense layer on sub-batches of this many images
    if (argc > 8) net.head_chunk = atoi(argv[8]);
    if (argc > 9) net.matrix_cores = atoi(argv[9]) != 0;
    auto x = rd(dir + "/net_in.u64");
    vector<ciphertext3D> imgs;
    for (int b = 0; b < batch; b++) imgs.push_back(ciphertext3D::fromHost(x.data(), 1, 1, 28, 28));
    ciphertext3D in = stackImages(imgs);
    if (!resident) {      // layer by layer, coefficient form at every boundary: dump each output for the per-layer digests
        ciphertext3D t = in;
        for (int i = 0; i < net.getNumLayers(); i++) {
            net.getLayer(i)->out_form = CRC_COEFF;
            t = net.getLayer(i)->forward(t);
            wr(dir + "/layer_" + to_string(i) + ".u64", t.toHost());
        }
        wr(dir + "/out.u64", t.toHost());
    } else {
        ciphertext3D out = net.forward(in);
        wr(dir + "/out.u64", out.toHost());
        for (double ms : net.last_layer_ms) fprintf(stderr, "%.3f,", ms);
        fprintf(stderr, "\n");
    }
    delParameters();
    return 0;
}

// net3 <model> <h5> <dir> <batch>: the three NTT-resident runs of tests/test_gpu_host_cpp.py's full-size cases from ONE built network (the encode + lift + NTT
// of 10^5 .. 10^6 plaintexts is most of a case's time): Network::forward as built (one image), after Network::fuse() (one image), and fused on `batch` images.
// Writes out_unfused.u64, out_fused.u64, out_fused_batch.u64
static int do_net3(int argc, char **argv)
{
    if (argc < 6) return 1;
    string model = argv[2], h5 = argv[3], dir = argv[4]; const int batch = atoi(argv[5]);
    setup(dir);
    ifstream evf(dir + "/evk.u64", ios::binary);
    if (evf) {
        auto evk = rd(dir + "/evk.u64");
        ev_keys16 = make_shared<DeviceBuffer>(evk.size() * 8);
        crc_memcpy_h2d(context, ev_keys16->ptr, evk.data(), evk.size() * 8, nullptr); crc_stream_sync(context, nullptr);
    }
    CnnBuilder builder(h5);
    Network net = builder.buildNetworkByName(model);
    net.ntt_resident = true;
    auto x = rd(dir + "/net_in.u64");
    const ciphertext3D one = ciphertext3D::fromHost(x.data(), 1, 1, 28, 28);
    // (the unfused run leaves the weights in the matrix-core forms; Network::fuse() rebuilds the canonical ones from the plaintexts before it folds)
    { ciphertext3D out = net.forward(one); wr(dir + "/out_unfused.u64", out.toHost()); }
    const int removed = net.fuse();
    fprintf(stderr, "fused: %d layers removed, %d left\n", removed, net.getNumLayers());
    { ciphertext3D out = net.forward(one); wr(dir + "/out_fused.u64", out.toHost()); }
    vector<ciphertext3D> imgs(batch, one);
    { ciphertext3D out = net.forward(stackImages(imgs)); wr(dir + "/out_fused_batch.u64", out.toHost()); }
    delParameters();
    return 0;
}

// netr <model> <h5> <dir> <batch> <layer_before_reenc> <fuse 0|1> <head_chunk>: the reference's published configurations -- Network::forward WITH the client-side
// refresh (network.cpp:30-34), now on the device (refreshImages).  <dir> holds params / evk / net_in as for `net` plus sk.u64 and pk.u64 (the client's keys).
// Writes pre_<i>.u64 for the layers in front of the refresh (layer by layer, coefficient form: the reference's digests; unfused runs only), reenc_floats.f32
// (the floats the client saw, per image), dec.u64 ([batch][10][n] decrypted output plaintexts), budget.u64, and prints the per-layer times with T_REENC
static int do_netr(int argc, char **argv)
{
    if (argc < 9) return 1;
    string model = argv[2], h5 = argv[3], dir = argv[4]; const int batch = atoi(argv[5]), reenc = atoi(argv[6]); const bool fuse = atoi(argv[7]) != 0;
    const int head_chunk = atoi(argv[8]);
    setDeterministicSeed(4242);
    setup(dir);
    secret_key = rd(dir + "/sk.u64"); public_key = rd(dir + "/pk.u64");
    { auto evk = rd(dir + "/evk.u64");
      ev_keys16 = make_shared<DeviceBuffer>(evk.size() * 8);
      crc_memcpy_h2d(context, ev_keys16->ptr, evk.data(), evk.size() * 8, nullptr); crc_stream_sync(context, nullptr); }
    CnnBuilder builder(h5);
    Network net = builder.buildNetworkByName(model);
    auto x = rd(dir + "/net_in.u64");
    const ciphertext3D one = ciphertext3D::fromHost(x.data(), 1, 1, 28, 28);
    if (!fuse) {
        ciphertext3D t = one;
        for (int i = 0; i < reenc; i++) { net.getLayer(i)->out_form = CRC_COEFF; t = net.getLayer(i)->forward(t); wr(dir + "/pre_" + to_string(i) + ".u64", t.toHost()); }
    }
    net.ntt_resident = true; net.layer_before_reenc = reenc; net.keep_reenc_values = true; net.head_chunk = head_chunk;
    if (fuse) { const int removed = net.fuse(); fprintf(stderr, "fused: %d layers removed, %d left, refresh in front of layer %d\n", removed, net.getNumLayers(),
        net.layer_before_reenc); }
    vector<ciphertext3D> imgs(batch, one);
    ciphertext3D out = net.forward(stackImages(imgs));
    { ofstream f(dir + "/reenc_floats.f32", ios::binary); f.write((const char *)net.last_reenc_values.data(), net.last_reenc_values.size() * 4); }
    vector<u64> h = out.toHost(), pl(out.count() * (size_t)crc_ctx_n(context)), bud;
    if (crc_decrypt(context, secret_key.data(), h.data(), out.count(), 2, pl.data())) return 3;
    for (size_t i = 0; i < out.count(); i++) bud.push_back((u64)noiseBudget(out, i));
    wr(dir + "/dec.u64", pl); wr(dir + "/budget.u64", bud);
    for (double ms : net.last_layer_ms) fprintf(stderr, "%.3f,", ms);
    fprintf(stderr, " T_REENC %.3f\n", net.last_reenc_ms);
    // a second forward draws fresh randomness: other ciphertexts, the same plaintexts
    ciphertext3D out2 = net.forward(stackImages(imgs));
    vector<u64> h2 = out2.toHost(), pl2(pl.size());
    if (crc_decrypt(context, secret_key.data(), h2.data(), out2.count(), 2, pl2.data())) return 3;
    if (h2 == h) { fprintf(stderr, "the refresh reused its randomness\n"); return 4; }
    if (pl2 != pl) { fprintf(stderr, "second forward decrypts differently\n"); return 5; }
    delParameters();
    printf("netr ok\n");
    return 0;
}

static vector<double> rdf(const string &p)
{
    ifstream f(p, ios::binary); if (!f) { fprintf(stderr, "missing %s\n", p.c_str()); exit(2); }
    f.seekg(0, ios::end); size_t sz = f.tellg(); f.seekg(0); vector<double> v(sz / 8); f.read((char *)v.data(), sz); return v;
}
static int do_files(int argc, char **argv)
{
    if (argc < 3) return 1;
    const string dir = argv[2];
    setup(dir);
    secret_key = rd(dir + "/sk.u64"); public_key = rd(dir + "/pk.u64");            // the fixture's key pair (globals are public, as in the reference)
    auto dims = rd(dir + "/layer_dims.u64");
    const int zd = (int)dims[0], xd = (int)dims[1], yd = (int)dims[2], xs = (int)dims[3], ys = (int)dims[4], xf = (int)dims[5], yf = (int)dims[6], nf =
        (int)dims[7], od = (int)dims[8];
    const int xo = (xd - xf) / xs + 1, yo = (yd - yf) / ys + 1;
    auto run = [&](Layer &c, Layer &b, Layer &f, const ciphertext3D &x, const string &out) {
        c.out_form = b.out_form = f.out_form = CRC_COEFF;
        wr(dir + "/" + out, f.forward(b.forward(c.forward(x))).toHost());
    };
    {   // what the reference wrote
        ifstream in(dir + "/ref_encoded_layers.bin", ios::binary); if (!in) { fprintf(stderr, "missing ref_encoded_layers.bin\n"); return 2; }
        ConvolutionalLayer c("conv", xd, yd, zd, xs, ys, xf, yf, nf, 2, &in);
        BatchNormLayer b("bn", nf, &in);
        FullyConnectedLayer f("fc", nf * xo * yo, od, 2, &in);
        if (in.peek() != EOF) { fprintf(stderr, "encoded-model stream not consumed exactly\n"); return 3; }
        run(c, b, f, loadEncryptedImage(zd, xd, yd, dir + "/ref_cipher_image.bin"), "out_from_ref_files.u64");
    }
    {   // the same files written by the host classes (encoding as CnnBuilder::build*Layer: float32 widened to double)
        auto enc = [&](const vector<double> &v) { vector<Plaintext> o; for (double d : v) o.push_back(fraencode((double)(float)d)); return o; };
        auto fw = enc(rdf(dir + "/conv_w.f64")), fb = enc(rdf(dir + "/conv_b.f64")), bm = enc(rdf(dir + "/bn_mean.f64")), dw = enc(rdf(dir + "/fc_w.f64")),
            db = enc(rdf(dir + "/fc_b.f64"));
        // cnnBuilder.cpp:100-102
        vector<Plaintext> bv; for (double d : rdf(dir + "/bn_var.f64")) { float v = (float)d; v = 1 / sqrt(v + 0.00001); bv.push_back(fraencode((double)v)); }
        plaintext4D ew(nf, plaintext3D(zd, plaintext2D(xf, vector<Plaintext>(yf)))); size_t w = 0;
        for (int n = 0; n < nf; n++) for (int z = 0; z < zd; z++) for (int i = 0; i < xf; i++) for (int j = 0; j < yf; j++) ew[n][z][i][j] = fw[w++];
        plaintext2D ed(od, vector<Plaintext>(nf * xo * yo)); w = 0;
        for (int i = 0; i < od; i++) for (int j = 0; j < nf * xo * yo; j++) ed[i][j] = dw[w++];
        ConvolutionalLayer c("conv", xd, yd, zd, xs, ys, xf, yf, nf, 2, ew, fb);
        BatchNormLayer b("bn", nf, bm, bv);
        FullyConnectedLayer f("fc", nf * xo * yo, od, 2, ed, db);
        { ofstream o(dir + "/our_encoded_layers.bin", ios::binary); c.savePlaintextParameters(&o); b.savePlaintextParameters(&o);
            f.savePlaintextParameters(&o); }
        vector<float> image; for (double d : rdf(dir + "/image.f64")) image.push_back((float)d);
        ciphertext3D x = encryptAndSaveImage(image, zd, xd, yd, dir + "/our_cipher_image.bin");
        run(c, b, f, x, "out_from_our_files.u64");
    }
    delParameters();
    printf("files ok\n");
    return 0;
}

static const char *status_name(exit_status_forward s) { return s == SUCCESS ? "SUCCESS" : s == OUT_OF_BUDGET ? "OUT_OF_BUDGET" : "MISPREDICTED"; }

static int do_searchlogic(int argc, char **argv)
{
    if (argc < 7) return 1;
    const u64 lo = strtoull(argv[2], 0, 0), hi = strtoull(argv[3], 0, 0), first_good = strtoull(argv[4], 0, 0), last_good = strtoull(argv[5], 0, 0), min_q =
        strtoull(argv[6], 0, 0);
    vector<pair<u64, exit_status_forward>> tried;
    auto pred = [&](u64 t) { exit_status_forward s = t < first_good ? MISPREDICTED : t > last_good ? OUT_OF_BUDGET : SUCCESS; tried.emplace_back(t, s);
        return s; };
    const u64 found = plainModulusBinarySearch(pred, lo, hi, min_q);
    printf("found %llu\n", (unsigned long long)found);
    for (auto &p : tried) printf("tried %llu %s\n", (unsigned long long)p.first, status_name(p.second));
    return 0;
}

static int do_search(int argc, char **argv)
{
    if (argc < 10) return 1;
    PlainModulusSearch s;
    s.model = argv[2];
    const string h5 = argv[3], images = argv[4];
    s.max_poly_modulus = atoi(argv[5]);
    const u64 lo = strtoull(argv[6], 0, 0), hi = strtoull(argv[7], 0, 0);
    const int num_images = atoi(argv[8]); s.seed = (unsigned)strtoul(argv[9], 0, 0);
    for (int i = 10; i < argc; i++) s.coeff_modulus.push_back(strtoull(argv[i], 0, 0));
    ifstream f(images, ios::binary); if (!f) { fprintf(stderr, "missing %s\n", images.c_str()); return 2; }
    f.seekg(0, ios::end); const size_t cnt = (size_t)f.tellg() / (784 * 4); f.seekg(0);
    s.test_set.assign(cnt, vector<float>(784));
    for (auto &im : s.test_set) f.read((char *)im.data(), 784 * 4);
    s.predictWithPlainModel(h5);
    for (size_t i = 0; i < cnt; i++) printf("label %zu %d\n", i, (int)s.predicted_labels[i]);
    const u64 found = s.run(num_images, lo, hi, h5);
    printf("found %llu\n", (unsigned long long)found);
    for (size_t i = 0; i < s.tried.size(); i++) printf("tried %llu %s %.2f\n", (unsigned long long)s.tried[i].first, status_name(s.tried[i].second),
        s.test_seconds[i]);
    return 0;
}

#define EXPECT_THROW(stmt, type) do { bool ok_ = false; try { stmt; } catch (const type &) { ok_ = true; } catch (...) {} if (!ok_) { fprintf(stderr, "expected " #type " from: " #stmt "\n"); return 3; } } while (0)

static int do_api(int argc, char **argv)
{
    if (argc < 4) return 1;
    string h5 = argv[2], dir = argv[3];
    EXPECT_THROW(fraencode(1.0), logic_error);                         // context not set
    EXPECT_THROW(setParameters(4095, 1 << 20), invalid_argument);      // not a power of two
    setParameters(1024, {0x7fffffff380001ULL, 0x3fffffff000001ULL}, 1ULL << 20, 0);
    // client side round trip (encryptImage / decryptImage, globals.cpp:127-157,207-230)
    vector<float> img(28 * 28); for (int i = 0; i < 784; i++) img[i] = (float)((i % 17) - 8) / 4.0f;
    ciphertext3D ct = encryptImage(img, 1, 28, 28);
    floatCube back = decryptImage(ct);
    for (int i = 0; i < 28; i++) for (int j = 0; j < 28; j++) if (fabs(back[0][i][j] - img[i * 28 + j]) > 1e-6) { fprintf(stderr, "decrypt mismatch\n");
        return 4; }
    if (noiseBudget(ct) < 20) { fprintf(stderr, "budget too small\n"); return 4; }
    // a tiny layer stack: conv -> avgpool -> square -> fc, resident vs layerwise must give identical ciphertexts
    vector<float> w(2 * 1 * 3 * 3), b(2), fw(3 * 2 * 6 * 6), fb(3);
    for (size_t i = 0; i < w.size(); i++) w[i] = 0.05f * (float)((int)(i % 7) - 3);
    b[0] = 0.1f; b[1] = -0.2f;
    for (size_t i = 0; i < fw.size(); i++) fw[i] = 0.01f * (float)((int)(i % 11) - 5);
    fb = {0.5f, -0.25f, 0.125f};
    auto enc = [&](float v) { return fraencode((double)v); };
    plaintext4D ew(2, plaintext3D(1, plaintext2D(3, vector<Plaintext>(3)))); vector<Plaintext> eb(2);
    for (int f = 0; f < 2; f++) { for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) ew[f][0][i][j] = enc(w[(f * 3 + i) * 3 + j]); eb[f] = enc(b[f]); }
    plaintext2D efw(3, vector<Plaintext>(72)); vector<Plaintext> efb(3);
    for (int i = 0; i < 3; i++) { for (int j = 0; j < 72; j++) efw[i][j] = enc(fw[i * 72 + j]); efb[i] = enc(fb[i]); }
    vector<float> small(14 * 14); for (int i = 0; i < 196; i++) small[i] = (float)((i * 7) % 13 - 6) / 8.0f;
    ciphertext3D x = encryptImage(small, 1, 14, 14);
    Network net;
    net.getLayers().push_back(shared_ptr<Layer>(new ConvolutionalLayer("c", 14, 14, 1, 1, 1, 3, 3, 2, 4, ew, eb)));
    net.getLayers().push_back(shared_ptr<Layer>(new AvgPoolingLayer("p", 12, 12, 2, 2, 2, 2, 2)));
    net.getLayers().push_back(shared_ptr<Layer>(new SquareLayer("s", 2)));
    net.getLayers().push_back(shared_ptr<Layer>(new FullyConnectedLayer("f", 72, 3, 2, efw, efb)));
    net.ntt_resident = true;  vector<u64> r1 = net.forward(x).toHost();
    net.ntt_resident = false; ciphertext3D o2 = net.forward(x); vector<u64> r2 = o2.toHost();
    if (r1 != r2) { fprintf(stderr, "resident and layerwise outputs differ\n"); return 5; }
    // semantic check against float arithmetic
    floatCube dec = decryptImage(o2);
    double conv[2][12][12], pool[2][6][6];
    for (int f = 0; f < 2; f++) for (int i = 0; i < 12; i++) for (int j = 0; j < 12; j++) {
        double s = b[f];
        for (int a = 0; a < 3; a++) for (int c = 0; c < 3; c++) s += (double)w[(f * 3 + a) * 3 + c] * small[(i + a) * 14 + j + c];
        conv[f][i][j] = s;
    }
    for (int f = 0; f < 2; f++) for (int i = 0; i < 6; i++) for (int j = 0; j < 6; j++) { double s = (conv[f][2*i][2*j] + conv[f][2*i][2*j+1] +
        conv[f][2*i+1][2*j] + conv[f][2*i+1][2*j+1]) / 4; pool[f][i][j] = s * s; }
    for (int o = 0; o < 3; o++) { double s = fb[o];
        for (int f = 0; f < 2; f++) for (int i = 0; i < 6; i++) for (int j = 0; j < 6; j++) s += (double)fw[o * 72 + (f * 6 + i) * 6 + j] * pool[f][i][j];
        if (fabs(s - dec[0][o][0]) > 1e-4) { fprintf(stderr, "semantic mismatch %d: %f vs %f\n", o, s, dec[0][o][0]); return 6; } }
    // refresh path (network.cpp:30-34) keeps the result
    net.layer_before_reenc = 3; floatCube dec2 = decryptImage(net.forward(x)); net.layer_before_reenc = -1;
    for (int o = 0; o < 3; o++) if (fabs(dec2[0][o][0] - dec[0][o][0]) > 1e-4) { fprintf(stderr, "refresh changed the result\n"); return 7; }
    // fuse() and the refresh point: the refresh stays in front of the SAME layer when folds below it shift the indices, and no fold spans it.
    // [conv, avgpool, square, pool2, fc] with the refresh in front of pool2 (index 3): conv + avgpool fold (index 3 -> 2), square + pool2 must not pair up.
    {
        plaintext2D gw(3, vector<Plaintext>(50)); vector<Plaintext> gb(3);
        for (int i = 0; i < 3; i++) { for (int j = 0; j < 50; j++) gw[i][j] = enc(fw[(i * 50 + j) % 216]); gb[i] = enc(fb[i]); }
        auto build = [&]() {
            Network nn;
            nn.getLayers().push_back(shared_ptr<Layer>(new ConvolutionalLayer("c", 14, 14, 1, 1, 1, 3, 3, 2, 4, ew, eb)));
            nn.getLayers().push_back(shared_ptr<Layer>(new AvgPoolingLayer("p", 12, 12, 2, 2, 2, 2, 2)));
            nn.getLayers().push_back(shared_ptr<Layer>(new SquareLayer("s", 2)));
            nn.getLayers().push_back(shared_ptr<Layer>(new PoolingLayer("p2", 6, 6, 2, 1, 1, 2, 2)));
            nn.getLayers().push_back(shared_ptr<Layer>(new FullyConnectedLayer("g", 50, 3, 2, gw, gb)));
            return nn;
        };
        Network plain = build(); plain.layer_before_reenc = 3;
        floatCube want = decryptImage(plain.forward(x));
        Network fusedn = build(); fusedn.layer_before_reenc = 3;
        const int removed = fusedn.fuse();
        const int at = fusedn.layer_before_reenc;
        if (at < 0 || at >= fusedn.getNumLayers() || fusedn.getLayer(at)->getName() != "p2") {
            fprintf(stderr, "fuse() moved the refresh point: %d layers removed, refresh now in front of index %d (%s)\n", removed, at,
                    at >= 0 && at < fusedn.getNumLayers() ? fusedn.getLayer(at)->getName().c_str() : "?");
            return 14;
        }
        if (at != 3 - removed) { fprintf(stderr, "refresh index %d after %d folds below it\n", at, removed); return 14; }
        floatCube got = decryptImage(fusedn.forward(x));
        for (int o = 0; o < 3; o++)
            if (fabs(got[0][o][0] - want[0][o][0]) > 1e-4) { fprintf(stderr, "fused network with a refresh differs: %f vs %f\n", got[0][o][0], want[0][o][0]);
                return 14; }
        // without a refresh the same network does pair square + pool2, bit-identically
        Network a = build(), b2 = build();
        const int removed2 = b2.fuse();
        if (removed2 <= removed) { fprintf(stderr, "square + pooling did not pair up without a refresh (%d vs %d)\n", removed2, removed); return 14; }
        if (a.forward(x).toHost() != b2.forward(x).toHost()) { fprintf(stderr, "fused network differs from the unfused one\n"); return 14; }
    }
    // save / load of the encoded parameters in SEAL's Plaintext wire format (savePlaintextParameters / istream ctor)
    { ofstream f(dir + "/enc_model.bin", ios::binary); for (int i = 0; i < net.getNumLayers(); i++) net.getLayer(i)->savePlaintextParameters(&f); }
    { ifstream f(dir + "/enc_model.bin", ios::binary);
      Network n2;
      n2.getLayers().push_back(shared_ptr<Layer>(new ConvolutionalLayer("c", 14, 14, 1, 1, 1, 3, 3, 2, 4, &f)));
      n2.getLayers().push_back(shared_ptr<Layer>(new AvgPoolingLayer("p", 12, 12, 2, 2, 2, 2, 2)));
      n2.getLayers().push_back(shared_ptr<Layer>(new SquareLayer("s", 2)));
      n2.getLayers().push_back(shared_ptr<Layer>(new FullyConnectedLayer("f", 72, 3, 2, &f)));
      if (n2.forward(x).toHost() != r1) { fprintf(stderr, "reloaded network differs\n"); return 8; } }
    // multi-GPU start-up path (Network::broadcastParameters over crc_comm / RCCL) on a one-rank communicator: the weights are unpacked,
    // "broadcast", checksummed and compared, and the network must still produce the same ciphertexts
    {
        uint8_t id[CRC_COMM_ID_BYTES]; crc_comm *comm = nullptr;
        if (crc_comm_unique_id(id) || crc_comm_create(context, 1, 0, id, &comm)) { fprintf(stderr, "crc_comm_create failed (rccl error %d)\n",
            crc_last_comm_error()); return 13; }
        const size_t bytes = net.broadcastParameters(comm, 0);
        fprintf(stderr, "broadcastParameters: %zu bytes on %d rank(s)\n", bytes, crc_comm_world(comm));
        net.ntt_resident = true;
        if (bytes == 0 || net.forward(x).toHost() != r1) { fprintf(stderr, "network differs after broadcastParameters\n"); return 13; }
        EXPECT_THROW(net.broadcastParameters(comm, 1), invalid_argument);
        crc_comm_destroy(comm);
    }
    // error behaviour mirrors the reference (std::invalid_argument on bad shapes / truncated streams)
    EXPECT_THROW(net.getLayer(0)->forward(ciphertext3D(1, 1, 10, 10)), invalid_argument);
    { istringstream empty(""); EXPECT_THROW(FullyConnectedLayer("f", 4, 2, 1, &empty), invalid_argument); }
    EXPECT_THROW(CnnBuilder("/nonexistent.h5").getPretrained("x"), runtime_error);
    // key / image files in SEAL's wire formats (setAndSaveParameters / initFromKeys / encryptAndSaveImage / loadEncryptedImage)
    {
        setDeterministicSeed(777);
        setAndSaveParameters(dir + "/pk.bin", dir + "/sk.bin", dir + "/evk.bin", 2048, 1ULL << 16);
        ciphertext3D saved = encryptAndSaveImage(small, 1, 14, 14, dir + "/img.bin");
        vector<u64> before = saved.toHost();
        setDeterministicSeed(999);                                    // a different key pair would be generated ...
        initFromKeys(dir + "/pk.bin", dir + "/sk.bin", dir + "/evk.bin", 2048, 1ULL << 16);       // ... but the files restore the first one
        ciphertext3D loaded = loadEncryptedImage(1, 14, 14, dir + "/img.bin");
        if (loaded.toHost() != before) { fprintf(stderr, "image file round trip differs\n"); return 11; }
        floatCube d3 = decryptImage(loaded);
        for (int i = 0; i < 196; i++) if (fabs(d3[0][i / 14][i % 14] - small[i]) > 1e-6) { fprintf(stderr, "decrypt after initFromKeys failed\n"); return 12; }
        EXPECT_THROW(initFromKeys(dir + "/pk.bin", dir + "/sk.bin", dir + "/evk.bin", 2048, 1ULL << 17), invalid_argument);   // hash mismatch
        setParameters(1024, {0x7fffffff380001ULL, 0x3fffffff000001ULL}, 1ULL << 20, 0);
    }
    // HDF5 loader through the builder
    CnnBuilder builder(h5);
    if (builder.getPretrained("pool1_features.conv1.weight").size() != 800) return 9;
    net.printNetworkStructure();
    delParameters();
    printf("api ok\n");
    return 0;
}

// test_host bcast <rank> <world> <rendezvous file> <out dir> [device]
//   Network::broadcastParameters across PROCESSES (one per rank; RCCL when every rank has its own GPU, the shared-memory rehearsal transport --
//   CRC_COMM_TRANSPORT=shm --
//   when they share one): rank 0 holds the real weights, every other rank builds the same topology from DIFFERENT weights, joins through the id rank 0 left in
//   the
//   file, receives -- and must then produce rank 0's output ciphertexts bit for bit (<out dir>/bcast_out_<rank>.u64)
static int do_bcast(int argc, char **argv)
{
    if (argc < 6) return 1;
    const int rank = atoi(argv[2]), world = atoi(argv[3]); const string rdv = argv[4], dir = argv[5]; const int device = argc > 6 ? atoi(argv[6]) : 0;
    setDeterministicSeed(4242);
    setParameters(1024, {0x7fffffff380001ULL, 0x3fffffff000001ULL}, 1ULL << 20, device);
    // the non-root ranks start from other weights: only the broadcast can make the outputs agree
    const float scale = rank == 0 ? 1.0f : -0.5f;
    auto enc = [&](float v) { return fraencode((double)(v * scale)); };
    plaintext4D ew(2, plaintext3D(1, plaintext2D(3, vector<Plaintext>(3)))); vector<Plaintext> eb(2);
    for (int f = 0; f < 2; f++) { for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) ew[f][0][i][j] = enc(0.05f * (float)(((f * 3 + i) * 3 + j) % 7 -
        3)); eb[f] = enc(f ? -0.2f : 0.1f); }
    plaintext2D efw(3, vector<Plaintext>(72)); vector<Plaintext> efb(3);
    for (int i = 0; i < 3; i++) { for (int j = 0; j < 72; j++) efw[i][j] = enc(0.01f * (float)((i * 72 + j) % 11 - 5)); efb[i] = enc(0.125f * (float)(i + 1)); }
    vector<float> small(14 * 14); for (int i = 0; i < 196; i++) small[i] = (float)((i * 7) % 13 - 6) / 8.0f;
    ciphertext3D x = encryptImage(small, 1, 14, 14);
    Network net;
    net.getLayers().push_back(shared_ptr<Layer>(new ConvolutionalLayer("c", 14, 14, 1, 1, 1, 3, 3, 2, 4, ew, eb)));
    net.getLayers().push_back(shared_ptr<Layer>(new AvgPoolingLayer("p", 12, 12, 2, 2, 2, 2, 2)));
    net.getLayers().push_back(shared_ptr<Layer>(new SquareLayer("s", 2)));
    net.getLayers().push_back(shared_ptr<Layer>(new FullyConnectedLayer("f", 72, 3, 2, efw, efb)));
    uint8_t id[CRC_COMM_ID_BYTES];
    if (rank == 0) {
        if (crc_comm_unique_id(id)) { fprintf(stderr, "crc_comm_unique_id failed (rccl error %d)\n", crc_last_comm_error()); return 13; }
        { ofstream o(rdv + ".tmp", ios::binary); o.write((const char *)id, sizeof id); }
        if (rename((rdv + ".tmp").c_str(), rdv.c_str())) return 13;
    } else {
        bool got = false;
        for (int tries = 0; tries < 1200 && !got; tries++) { ifstream f(rdv, ios::binary); got = f && f.read((char *)id, sizeof id); if (!got) usleep(100000); }
        if (!got) { fprintf(stderr, "no rendezvous id\n"); return 13; }
    }
    crc_comm *comm = nullptr;
    if (crc_comm_create(context, world, rank, id, &comm)) { fprintf(stderr, "crc_comm_create failed (rccl error %d)\n", crc_last_comm_error()); return 13; }
    const size_t bytes = net.broadcastParameters(comm, 0);
    net.ntt_resident = true;
    wr(dir + "/bcast_out_" + to_string(rank) + ".u64", net.forward(x).toHost());
    printf("bcast ok: rank %d of %d, %zu bytes\n", rank, crc_comm_world(comm), bytes);
    crc_comm_destroy(comm);
    delParameters();
    return 0;
}

int main(int argc, char **argv)
{
    if (argc < 2) return 1;
    try {
        if (!strcmp(argv[1], "bcast")) return do_bcast(argc, argv);
        if (!strcmp(argv[1], "net")) return do_net(argc, argv);
        if (!strcmp(argv[1], "net3")) return do_net3(argc, argv);
        if (!strcmp(argv[1], "netr")) return do_netr(argc, argv);
        if (!strcmp(argv[1], "api")) return do_api(argc, argv);
        if (!strcmp(argv[1], "files")) return do_files(argc, argv);
        if (!strcmp(argv[1], "searchlogic")) return do_searchlogic(argc, argv);
        if (!strcmp(argv[1], "search")) return do_search(argc, argv);
    } catch (const exception &e) { fprintf(stderr, "exception: %s\n", e.what()); return 10; }
    return 1;
}
