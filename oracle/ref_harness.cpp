// ref_harness.cpp -- TEST INFRASTRUCTURE.  Our own driver, linked against the *reference itself* (SEAL 2.3.1 and the
// CrCNN layer sources compiled in place from /root/reference by oracle/Makefile).  It loads keys / ciphertexts /
// plaintexts produced by our oracle (raw little-endian uint64 files, layout [..][k][n] without SEAL's pad word),
// pushes them through the reference's Evaluator and Layer classes, and writes the reference's outputs back as raw
// files.  oracle/make_golden.py packs those into tests/golden/.  Never shipped, never on the product path.
//
//   ref_harness ops <dir>    op-level vectors (+ two-way key/ciphertext compatibility checks)
//   ref_harness layers <dir> CrCNN layer-level vectors on a small tensor
//   ref_harness net <dir>    a whole CrCNN network (topology + float weights from <dir>), SHA-256 per layer
//   ref_harness files <dir>  CrCNN's own file formats: the reference WRITES an encoded-model stream (savePlaintextParameters of a conv, a batch-norm
//                            and a dense layer back to back: cnnBuilder.cpp:181-196) and a cipher_image file (encryptAndSaveImage, globals.cpp:174-190),
//                            runs the layers on that image, and -- if <dir> holds our_encoded_layers.bin / our_cipher_image.bin written by the
//                            product -- LOADS those with its istream constructors / loadEncryptedImage (globals.cpp:193-205) and runs them too
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>
#include <fstream>
#include <iostream>
#include <sstream>
#include <memory>
#include <chrono>
#include <thread>
#include <mutex>
#include <random>
#include <map>
#include <cmath>
#include <algorithm>
#include <functional>
#include <limits>
#include <array>
#include <set>
#include <unordered_map>
#include <stdexcept>
#include <iomanip>
#include <atomic>
#include <shared_mutex>
#include <utility>
#include <type_traits>

// the harness needs to fill Ciphertext / key containers from raw buffers; SEAL keeps resize() and hash_block_ private
#define private public
#define protected public
#include "seal/seal.h"
#undef private
#undef protected
#include "globals.h"
#include "layer.h"
#include "network.h"
#include "convolutionalLayer.h"
#include "fullyConnectedLayer.h"
#include "poolingLayer.h"
#include "avgPoolingLayer.h"
#include "squareLayer.h"
#include "batchNormLayer.h"

using namespace std;
using namespace seal;
typedef uint64_t u64;

static string DIR;
static vector<u64> rd(const string &name, bool must = true)
{
    ifstream f(DIR + "/" + name, ios::binary);
    if (!f) { if (must) { fprintf(stderr, "missing %s\n", name.c_str()); exit(2); } return {}; }
    f.seekg(0, ios::end); size_t sz = f.tellg(); f.seekg(0);
    vector<u64> v(sz / 8); f.read((char *)v.data(), sz); return v;
}
static void wr(const string &name, const vector<u64> &v)
{
    ofstream f(DIR + "/" + name, ios::binary); f.write((const char *)v.data(), v.size() * 8);
}
static vector<double> rdf(const string &name)
{
    ifstream f(DIR + "/" + name, ios::binary); f.seekg(0, ios::end); size_t sz = f.tellg(); f.seekg(0);
    vector<double> v(sz / 8); f.read((char *)v.data(), sz); return v;
}

// deterministic RNG for the reference's own keygen/encrypt (default is std::random_device, randomgen.cpp:7)
struct DetRng : UniformRandomGenerator { mt19937_64 g; DetRng(u64 s) : g(s) {} uint32_t generate() override { return (uint32_t)g(); } };
struct DetFactory : UniformRandomGeneratorFactory { u64 ctr = 1; UniformRandomGenerator *create() override { return new DetRng(0xC0FFEE + ctr++); } };
static DetFactory det_factory;

static int N, K; static u64 T;
static vector<u64> Q;

static void setup()
{
    auto p = rd("params.u64");          // n, k, t, q[0..k)
    N = (int)p[0]; K = (int)p[1]; T = p[2]; Q.assign(p.begin() + 3, p.begin() + 3 + K);
    parms = new EncryptionParameters();
    parms->set_poly_modulus("1x^" + to_string(N) + " + 1");
    vector<SmallModulus> mods; for (u64 q : Q) mods.emplace_back(q);
    parms->set_coeff_modulus(mods);
    parms->set_plain_modulus(T);
    parms->set_random_generator(&det_factory);
    context = new SEALContext(*parms);
    if (!context->qualifiers().parameters_set || !context->qualifiers().enable_ntt) { fprintf(stderr, "bad parameters\n"); exit(3); }
    evaluator = new Evaluator(*context);
    fraencoder = new FractionalEncoder(context->plain_modulus(), context->poly_modulus(), 64, 32, 3);
}

// ---- conversions between our [..][k][n] layout and SEAL's [..][k][n+1] ----
static Ciphertext to_ct(const u64 *src, int size)
{
    Ciphertext ct(*parms, size);
    ct.resize(*parms, size);
    for (int p = 0; p < size; p++) for (int i = 0; i < K; i++) {
        u64 *d = ct.data(p) + (size_t)i * (N + 1);
        memcpy(d, src + ((size_t)p * K + i) * N, 8 * (size_t)N); d[N] = 0;
    }
    return ct;
}
static void from_ct(const Ciphertext &ct, vector<u64> &out)
{
    int size = ct.size();
    for (int p = 0; p < size; p++) for (int i = 0; i < K; i++) {
        const u64 *s = ct.data(p) + (size_t)i * (N + 1);
        if (s[N] != 0) { fprintf(stderr, "pad word not zero\n"); exit(4); }
        out.insert(out.end(), s, s + N);
    }
}
static Plaintext to_plain(const u64 *src, int cc)
{
    Plaintext p(cc);
    for (int i = 0; i < cc && i < N; i++) p[i] = src[i];
    return p;
}
static void from_plain(const Plaintext &p, vector<u64> &out, int words)
{
    for (int i = 0; i < words; i++) out.push_back(i < p.coeff_count() ? p[i] : 0);
}
static void from_plain_ntt(const Plaintext &p, vector<u64> &out)   // [k][n+1] -> [k][n]
{
    for (int i = 0; i < K; i++) { const u64 *s = p.data() + (size_t)i * (N + 1); out.insert(out.end(), s, s + N); }
}

static SecretKey load_sk(const vector<u64> &v)
{
    SecretKey sk; sk.data().resize(N + 1, K * 64); sk.data().set_zero();
    for (int i = 0; i < K; i++) memcpy(sk.data().data() + (size_t)i * (N + 1), v.data() + (size_t)i * N, 8 * (size_t)N);
    sk.hash_block() = parms->hash_block();
    return sk;
}
static PublicKey load_pk(const vector<u64> &v)
{
    PublicKey pk; pk.data().resize(2, N + 1, K * 64); pk.data().set_zero();
    for (int p = 0; p < 2; p++) for (int i = 0; i < K; i++)
        memcpy(pk.data().data(p) + (size_t)i * (N + 1), v.data() + ((size_t)p * K + i) * N, 8 * (size_t)N);
    pk.hash_block() = parms->hash_block();
    return pk;
}
static int digits(u64 q, int dbc) { int L = 0; while (q) { L++; q >>= dbc; } return L; }
static void load_evk(const vector<u64> &v, int dbc, EvaluationKeys &ek)
{
    ek.data().clear(); ek.data().resize(1);
    const u64 *src = v.data();
    for (int l = 0; l < K; l++) {
        int L = digits(Q[l], dbc);
        ek.data()[0].emplace_back(to_ct(src, 2 * L));
        src += (size_t)2 * L * K * N;
    }
    ek.decomposition_bit_count_ = dbc;
    ek.hash_block() = parms->hash_block();
}
static void dump_evk(const EvaluationKeys &ek, vector<u64> &out)
{
    for (int l = 0; l < K; l++) from_ct(ek.data()[0][l], out);
}

// ---------------------------------------------------------------------------------------------------------------
static int do_ops()
{
    setup();
    auto skv = rd("sk.u64"), pkv = rd("pk.u64"), evkv = rd("evk.u64");
    auto cts = rd("ct_in.u64"), plains = rd("plains.u64"), pcc = rd("plain_cc.u64");
    size_t ctw = (size_t)2 * K * N;
    int nct = (int)(cts.size() / ctw), npl = (int)pcc.size();
    SecretKey sk = load_sk(skv); PublicKey pk = load_pk(pkv);
    Decryptor dec(*context, sk); Encryptor enc(*context, pk);
    EvaluationKeys ek; load_evk(evkv, 16, ek);

    // constants the reference derived for these parameters
    {
        vector<u64> c;
        for (int i = 0; i < K; i++) c.push_back(context->small_ntt_tables_[i].get_root());
        for (int i = 0; i < K; i++) { c.push_back(parms->coeff_modulus()[i].const_ratio()[0]); c.push_back(parms->coeff_modulus()[i].const_ratio()[1]); }
        for (int i = 0; i < K; i++) c.push_back(evaluator->coeff_div_plain_modulus_[i]);
        for (int i = 0; i < K; i++) c.push_back(evaluator->upper_half_increment_[i]);
        c.push_back(context->base_converter_.bsk_base_mod_count());
        for (auto &m : context->base_converter_.bsk_base_array_) c.push_back(m.value());
        for (auto &t : context->base_converter_.bsk_small_ntt_table_) c.push_back(t.get_root());
        wr("ref_consts.u64", c);
        vector<u64> rp;
        for (int i = 0; i < N; i++) rp.push_back(context->small_ntt_tables_[0].get_from_root_powers(i));
        for (int i = 0; i < N; i++) rp.push_back(context->small_ntt_tables_[0].get_from_inv_root_powers_div_two(i));
        wr("ref_root_powers0.u64", rp);
    }
    // (0) SEAL's own wire bytes of these objects (ciphertext / evaluation keys / public key / secret key), incl. the parameter hash
    {
        auto dump = [&](const string &name, const string &bytes) { ofstream f(DIR + "/" + name, ios::binary); f.write(bytes.data(), bytes.size()); };
        { ostringstream o; to_ct(&cts[0], 2).save(o); dump("ref_wire_ct.bin", o.str()); }
        { ostringstream o; ek.save(o); dump("ref_wire_evk.bin", o.str()); }
        { ostringstream o; pk.save(o); dump("ref_wire_pk.bin", o.str()); }
        { ostringstream o; sk.save(o); dump("ref_wire_sk.bin", o.str()); }
        vector<u64> h(parms->hash_block().begin(), parms->hash_block().end()); wr("ref_params_hash.u64", h);
    }
    // (1) the reference decrypts the oracle's ciphertexts under the oracle's secret key
    {
        vector<u64> out, bud;
        for (int i = 0; i < nct; i++) { Ciphertext ct = to_ct(&cts[i * ctw], 2); Plaintext p; dec.decrypt(ct, p); from_plain(p, out, N); bud.push_back(dec.invariant_noise_budget(ct)); }
        wr("ref_dec_in.u64", out); wr("ref_budget_in.u64", bud);
    }
    // (2) the reference encrypts under the oracle's public key (deterministic RNG) -> oracle must decrypt it
    {
        vector<u64> out;
        for (int j = 0; j < npl; j++) { Ciphertext ct; enc.encrypt(to_plain(&plains[(size_t)j * N], (int)pcc[j]), ct); from_ct(ct, out); }
        wr("ref_enc.u64", out);
    }
    // (3) reference keygen + evaluation keys, so the oracle's relinearize/decrypt are also exercised on SEAL-made keys
    {
        KeyGenerator kg(*context);
        EvaluationKeys ek2; kg.generate_evaluation_keys(16, ek2);
        vector<u64> s, p, e;
        for (int i = 0; i < K; i++) { const u64 *x = kg.secret_key().data().data() + (size_t)i * (N + 1); s.insert(s.end(), x, x + N); }
        for (int q = 0; q < 2; q++) for (int i = 0; i < K; i++) { const u64 *x = kg.public_key().data().data(q) + (size_t)i * (N + 1); p.insert(p.end(), x, x + N); }
        dump_evk(ek2, e);
        wr("ref_sk.u64", s); wr("ref_pk.u64", p); wr("ref_evk.u64", e);
        Encryptor enc2(*context, kg.public_key()); Decryptor dec2(*context, kg.secret_key());
        vector<u64> out, sq, rl, dsq;
        for (int j = 0; j < npl; j++) {
            Ciphertext ct; enc2.encrypt(to_plain(&plains[(size_t)j * N], (int)pcc[j]), ct); from_ct(ct, out);
            evaluator->square(ct); from_ct(ct, sq);
            evaluator->relinearize(ct, ek2); from_ct(ct, rl);
            Plaintext pp; dec2.decrypt(ct, pp); from_plain(pp, dsq, N);
        }
        wr("ref_enc2.u64", out); wr("ref_sq2.u64", sq); wr("ref_relin2.u64", rl); wr("ref_dec_relin2.u64", dsq);
    }
    // (4) evaluator ops on the oracle's ciphertexts
    vector<u64> o_ctntt, o_plntt, o_mulntt, o_mul, o_add, o_addp, o_subp, o_mulp, o_sq, o_rl, o_bud, o_dec;
    for (int i = 0; i < nct; i++) {
        Ciphertext ct = to_ct(&cts[i * ctw], 2);
        Ciphertext a(ct); evaluator->transform_to_ntt(a); from_ct(a, o_ctntt);
        for (int j = 0; j < npl; j++) {
            Plaintext pl = to_plain(&plains[(size_t)j * N], (int)pcc[j]);
            Plaintext pn(pl); evaluator->transform_to_ntt(pn);
            if (i == 0) from_plain_ntt(pn, o_plntt);
            Ciphertext m(a); evaluator->multiply_plain_ntt(m, pn); from_ct(m, o_mulntt);
            evaluator->transform_from_ntt(m); from_ct(m, o_mul);
            Ciphertext b(ct); evaluator->add_plain(b, pl); from_ct(b, o_addp);
            Ciphertext c(ct); evaluator->sub_plain(c, pl); from_ct(c, o_subp);
            Ciphertext d(ct); evaluator->multiply_plain(d, pl); from_ct(d, o_mulp);
        }
        Ciphertext e(ct); evaluator->add(e, to_ct(&cts[((i + 1) % nct) * ctw], 2)); from_ct(e, o_add);
        Ciphertext s(ct); evaluator->square(s); from_ct(s, o_sq);
        evaluator->relinearize(s, ek); from_ct(s, o_rl);
        o_bud.push_back(dec.invariant_noise_budget(s));
        Plaintext pp; dec.decrypt(s, pp); from_plain(pp, o_dec, N);
    }
    wr("ref_ct_ntt.u64", o_ctntt); wr("ref_plain_ntt.u64", o_plntt); wr("ref_mul_ntt.u64", o_mulntt); wr("ref_mul.u64", o_mul);
    wr("ref_add.u64", o_add); wr("ref_add_plain.u64", o_addp); wr("ref_sub_plain.u64", o_subp); wr("ref_mul_plain.u64", o_mulp);
    wr("ref_sq.u64", o_sq); wr("ref_relin.u64", o_rl); wr("ref_budget_relin.u64", o_bud); wr("ref_dec_relin.u64", o_dec);
    // (5) encoder
    {
        auto fl = rdf("floats.f64");
        vector<u64> enc_out, cc; vector<u64> decd;
        for (double v : fl) {
            Plaintext p = fraencoder->encode(v);
            cc.push_back(p.coeff_count()); from_plain(p, enc_out, N);
            double back = fraencoder->decode(p); u64 bits; memcpy(&bits, &back, 8); decd.push_back(bits);
        }
        wr("ref_enc_floats.u64", enc_out); wr("ref_enc_cc.u64", cc); wr("ref_decode.u64", decd);
    }
    return 0;
}

// ---------------------------------------------------------------------------------------------------------------
static ciphertext3D to_tensor(const vector<u64> &v, int zd, int xd, int yd)
{
    size_t ctw = (size_t)2 * K * N;
    ciphertext3D t(zd, ciphertext2D(xd, vector<Ciphertext>(yd)));
    for (int z = 0; z < zd; z++) for (int x = 0; x < xd; x++) for (int y = 0; y < yd; y++)
        t[z][x][y] = to_ct(&v[(((size_t)z * xd + x) * yd + y) * ctw], 2);
    return t;
}
static void from_tensor(const ciphertext3D &t, vector<u64> &out)
{
    for (auto &a : t) for (auto &b : a) for (auto &c : b) from_ct(c, out);
}

static ConvolutionalLayer *make_conv(const string &name, int xd, int yd, int zd, int xs, int ys, int xf, int yf, int nf, int th,
                                     const vector<double> &w, const vector<double> &b)
{   // same encoding loop as CnnBuilder::buildConvolutionalLayer (cnnBuilder.cpp:25-50); floats are float32 widened
    plaintext4D ew(nf, plaintext3D(zd, plaintext2D(xf, vector<Plaintext>(yf)))); vector<Plaintext> eb(nf);
    size_t idx = 0;
    for (int n = 0; n < nf; n++) { for (int z = 0; z < zd; z++) for (int i = 0; i < xf; i++) for (int j = 0; j < yf; j++) ew[n][z][i][j] = fraencoder->encode((float)w[idx++]);
        eb[n] = fraencoder->encode((float)b[n]); }
    return new ConvolutionalLayer(name, xd, yd, zd, xs, ys, xf, yf, nf, th, ew, eb);
}
static FullyConnectedLayer *make_fc(const string &name, int in_dim, int out_dim, int th, const vector<double> &w, const vector<double> &b)
{   // cnnBuilder.cpp:53-76
    plaintext2D ew(out_dim, vector<Plaintext>(in_dim)); vector<Plaintext> eb(out_dim);
    size_t idx = 0;
    for (int i = 0; i < out_dim; i++) { for (int j = 0; j < in_dim; j++) ew[i][j] = fraencoder->encode((float)w[idx++]); eb[i] = fraencoder->encode((float)b[i]); }
    return new FullyConnectedLayer(name, in_dim, out_dim, th, ew, eb);
}
static BatchNormLayer *make_bn(const string &name, int ch, const vector<double> &mean, const vector<double> &var)
{   // cnnBuilder.cpp:89-105  (var[i] is a float; 1/sqrt(var+0.00001) evaluated in double then narrowed to float)
    vector<Plaintext> em(ch), ev(ch);
    for (int i = 0; i < ch; i++) { em[i] = fraencoder->encode((float)mean[i]); float v = (float)var[i]; v = 1 / sqrt(v + 0.00001); ev[i] = fraencoder->encode(v); }
    return new BatchNormLayer(name, ch, em, ev);
}

static int do_layers()
{
    setup();
    auto evkv = rd("evk.u64");
    ev_keys16 = new EvaluationKeys(); load_evk(evkv, 16, *ev_keys16);
    auto dims = rd("layer_dims.u64");   // zd xd yd | conv: xs ys xf yf nf | fc: out_dim | pool: xs ys xf yf
    int zd = dims[0], xd = dims[1], yd = dims[2];
    auto x = rd("layer_in.u64");
    auto fw = rdf("conv_w.f64"), fb = rdf("conv_b.f64");
    vector<u64> out;
    {   ConvolutionalLayer *l = make_conv("conv", xd, yd, zd, dims[3], dims[4], dims[5], dims[6], dims[7], 2, fw, fb);
        out.clear(); from_tensor(l->forward(to_tensor(x, zd, xd, yd)), out); wr("ref_conv.u64", out); delete l; }
    if (dims.size() > 13 && dims[13]) return 0;          // convolution only (one-channel set: the reference's dense layer does not survive zd = 1 with two threads)
    {   auto w = rdf("fc_w.f64"), b = rdf("fc_b.f64"); int od = dims[8];
        FullyConnectedLayer *l = make_fc("fc", zd * xd * yd, od, 2, w, b);
        out.clear(); from_tensor(l->forward(to_tensor(x, zd, xd, yd)), out); wr("ref_fc.u64", out); delete l; }
    {   PoolingLayer l("pool", xd, yd, zd, dims[9], dims[10], dims[11], dims[12]);
        out.clear(); from_tensor(l.forward(to_tensor(x, zd, xd, yd)), out); wr("ref_pool.u64", out); }
    {   AvgPoolingLayer l("avg", xd, yd, zd, dims[9], dims[10], dims[11], dims[12]);
        out.clear(); from_tensor(l.forward(to_tensor(x, zd, xd, yd)), out); wr("ref_avgpool.u64", out); }
    {   auto m = rdf("bn_mean.f64"), v = rdf("bn_var.f64");
        BatchNormLayer *l = make_bn("bn", zd, m, v);
        out.clear(); from_tensor(l->forward(to_tensor(x, zd, xd, yd)), out); wr("ref_bn.u64", out); delete l; }
    {   SquareLayer l("sq", 2);
        out.clear(); from_tensor(l.forward(to_tensor(x, zd, xd, yd)), out); wr("ref_square.u64", out); }
    return 0;
}

static int do_files()
{
    setup();
    auto pkv = rd("pk.u64"), skv = rd("sk.u64");
    PublicKey pk = load_pk(pkv); SecretKey sk = load_sk(skv);
    encryptor = new Encryptor(*context, pk); decryptor = new Decryptor(*context, sk);
    auto dims = rd("layer_dims.u64");   // zd xd yd | conv: xs ys xf yf nf | fc: out_dim
    const int zd = dims[0], xd = dims[1], yd = dims[2], nf = dims[7];
    const int xo = (xd - (int)dims[5]) / (int)dims[3] + 1, yo = (yd - (int)dims[6]) / (int)dims[4] + 1;
    auto fw = rdf("conv_w.f64"), fb = rdf("conv_b.f64"), bm = rdf("bn_mean.f64"), bv = rdf("bn_var.f64"), dw = rdf("fc_w.f64"), db = rdf("fc_b.f64");
    auto imgd = rdf("image.f64");
    vector<float> image(imgd.begin(), imgd.end());
    auto run = [&](ConvolutionalLayer *c, BatchNormLayer *b, FullyConnectedLayer *f, ciphertext3D x, const string &tag) {
        vector<u64> out; ciphertext3D y = f->forward(b->forward(c->forward(x)));
        from_tensor(y, out); wr("ref_files_out_" + tag + ".u64", out);
        vector<double> dec; floatCube img = decryptImage(y);
        for (auto &a : img) for (auto &r : a) for (float v : r) dec.push_back(v);
        ofstream o(DIR + "/ref_files_dec_" + tag + ".f64", ios::binary); o.write((const char *)dec.data(), dec.size() * 8);
    };
    {   // the reference writes ...
        ConvolutionalLayer *c = make_conv("conv", xd, yd, zd, dims[3], dims[4], dims[5], dims[6], nf, 2, fw, fb);
        BatchNormLayer *b = make_bn("bn", nf, bm, bv);
        FullyConnectedLayer *f = make_fc("fc", nf * xo * yo, dims[8], 2, dw, db);
        { ofstream o(DIR + "/ref_encoded_layers.bin", ofstream::binary); c->savePlaintextParameters(&o); b->savePlaintextParameters(&o); f->savePlaintextParameters(&o); }
        ciphertext3D x = encryptAndSaveImage(image, zd, xd, yd, DIR + "/ref_cipher_image.bin");
        run(c, b, f, x, "own");
        delete c; delete b; delete f;
    }
    ifstream ours(DIR + "/our_encoded_layers.bin", ifstream::binary);
    if (ours) {   // ... and reads what the product wrote
        ConvolutionalLayer c("conv", xd, yd, zd, dims[3], dims[4], dims[5], dims[6], nf, 2, &ours);
        BatchNormLayer b("bn", nf, &ours);
        FullyConnectedLayer f("fc", nf * xo * yo, dims[8], 2, &ours);
        ciphertext3D x = loadEncryptedImage(zd, xd, yd, DIR + "/our_cipher_image.bin");
        run(&c, &b, &f, x, "ours");
    }
    return 0;
}

// ---- SHA-256 (FIPS 180-4), for per-layer digests of full networks ----
struct Sha256 {
    uint32_t h[8]; uint8_t buf[64]; size_t fill = 0; uint64_t total = 0;
    Sha256() { static const uint32_t i[8] = {0x6a09e667,0xbb67ae85,0x3c6ef372,0xa54ff53a,0x510e527f,0x9b05688c,0x1f83d9ab,0x5be0cd19}; memcpy(h, i, 32); }
    static uint32_t rr(uint32_t x, int n) { return (x >> n) | (x << (32 - n)); }
    void block(const uint8_t *p) {
        static const uint32_t k[64] = {
        0x428a2f98,0x71374491,0xb5c0fbcf,0xe9b5dba5,0x3956c25b,0x59f111f1,0x923f82a4,0xab1c5ed5,0xd807aa98,0x12835b01,0x243185be,0x550c7dc3,0x72be5d74,0x80deb1fe,0x9bdc06a7,0xc19bf174,
        0xe49b69c1,0xefbe4786,0x0fc19dc6,0x240ca1cc,0x2de92c6f,0x4a7484aa,0x5cb0a9dc,0x76f988da,0x983e5152,0xa831c66d,0xb00327c8,0xbf597fc7,0xc6e00bf3,0xd5a79147,0x06ca6351,0x14292967,
        0x27b70a85,0x2e1b2138,0x4d2c6dfc,0x53380d13,0x650a7354,0x766a0abb,0x81c2c92e,0x92722c85,0xa2bfe8a1,0xa81a664b,0xc24b8b70,0xc76c51a3,0xd192e819,0xd6990624,0xf40e3585,0x106aa070,
        0x19a4c116,0x1e376c08,0x2748774c,0x34b0bcb5,0x391c0cb3,0x4ed8aa4a,0x5b9cca4f,0x682e6ff3,0x748f82ee,0x78a5636f,0x84c87814,0x8cc70208,0x90befffa,0xa4506ceb,0xbef9a3f7,0xc67178f2};
        uint32_t w[64];
        for (int i = 0; i < 16; i++) w[i] = (uint32_t)p[4*i] << 24 | (uint32_t)p[4*i+1] << 16 | (uint32_t)p[4*i+2] << 8 | p[4*i+3];
        for (int i = 16; i < 64; i++) { uint32_t s0 = rr(w[i-15],7) ^ rr(w[i-15],18) ^ (w[i-15] >> 3), s1 = rr(w[i-2],17) ^ rr(w[i-2],19) ^ (w[i-2] >> 10); w[i] = w[i-16] + s0 + w[i-7] + s1; }
        uint32_t a=h[0],b=h[1],c=h[2],d=h[3],e=h[4],f=h[5],g=h[6],hh=h[7];
        for (int i = 0; i < 64; i++) { uint32_t S1 = rr(e,6)^rr(e,11)^rr(e,25), ch = (e&f)^(~e&g), t1 = hh+S1+ch+k[i]+w[i], S0 = rr(a,2)^rr(a,13)^rr(a,22), mj = (a&b)^(a&c)^(b&c), t2 = S0+mj;
            hh=g; g=f; f=e; e=d+t1; d=c; c=b; b=a; a=t1+t2; }
        h[0]+=a;h[1]+=b;h[2]+=c;h[3]+=d;h[4]+=e;h[5]+=f;h[6]+=g;h[7]+=hh;
    }
    void update(const void *data, size_t len) { const uint8_t *p = (const uint8_t *)data; total += len;
        while (len) { size_t t = min(len, 64 - fill); memcpy(buf + fill, p, t); fill += t; p += t; len -= t; if (fill == 64) { block(buf); fill = 0; } } }
    string hex() { uint64_t bits = total * 8; uint8_t pad = 0x80; update(&pad, 1); uint8_t z = 0; while (fill != 56) update(&z, 1);
        uint8_t l[8]; for (int i = 0; i < 8; i++) l[i] = (uint8_t)(bits >> (56 - 8*i)); update(l, 8);
        char s[65]; for (int i = 0; i < 8; i++) sprintf(s + 8*i, "%08x", h[i]); return string(s); }
};
static string digest(const ciphertext3D &t)
{   // SHA-256 over the little-endian uint64 stream in [z][x][y][poly][k][n] order (pad words skipped)
    Sha256 s;
    for (auto &a : t) for (auto &b : a) for (auto &c : b) for (int p = 0; p < c.size(); p++) for (int i = 0; i < K; i++) s.update(c.data(p) + (size_t)i * (N + 1), 8 * (size_t)N);
    return s.hex();
}

// whole network: topology file lines "<kind> <name> <ints...>", weights as float64 files named <name>.<param>.f64
static int do_net()
{
    setup();
    auto evkv = rd("evk.u64", false);
    if (!evkv.empty()) { ev_keys16 = new EvaluationKeys(); load_evk(evkv, 16, *ev_keys16); }
    ifstream topo(DIR + "/topology.txt");
    struct Sl { string name; int in, out, th, slices; };
    map<int, Sl> sliced;
    Network net; string kind, name;
    while (topo >> kind >> name) {
        if (kind == "conv") { int xd, yd, zd, xs, ys, xf, yf, nf, th; topo >> xd >> yd >> zd >> xs >> ys >> xf >> yf >> nf >> th;
            net.getLayers().push_back(shared_ptr<Layer>(make_conv(name, xd, yd, zd, xs, ys, xf, yf, nf, th, rdf(name + ".weight.f64"), rdf(name + ".bias.f64")))); }
        else if (kind == "fc") { int in, out, th; topo >> in >> out >> th;
            net.getLayers().push_back(shared_ptr<Layer>(make_fc(name, in, out, th, rdf(name + ".weight.f64"), rdf(name + ".bias.f64")))); }
        else if (kind == "fcs") {   // the same FullyConnectedLayer, built and run in row slices so that the reference's k*(n+1)-word
                                    // NTT-form weight Plaintexts (fullyConnectedLayer.cpp:129-131) fit in host RAM; rows are independent
            int in, out, th, slices; topo >> in >> out >> th >> slices;
            sliced[(int)net.getLayers().size()] = {name, in, out, th, slices};
            net.getLayers().push_back(shared_ptr<Layer>(nullptr)); }
        else if (kind == "pool" || kind == "avgpool") { int xd, yd, zd, xs, ys, xf, yf; topo >> xd >> yd >> zd >> xs >> ys >> xf >> yf;
            if (kind == "pool") net.getLayers().push_back(shared_ptr<Layer>(new PoolingLayer(name, xd, yd, zd, xs, ys, xf, yf)));
            else net.getLayers().push_back(shared_ptr<Layer>(new AvgPoolingLayer(name, xd, yd, zd, xs, ys, xf, yf))); }
        else if (kind == "bn") { int ch; topo >> ch;
            net.getLayers().push_back(shared_ptr<Layer>(make_bn(name, ch, rdf(name + ".running_mean.f64"), rdf(name + ".running_var.f64")))); }
        else if (kind == "square") { int th; topo >> th; net.getLayers().push_back(shared_ptr<Layer>(new SquareLayer(name, th))); }
        else { fprintf(stderr, "unknown layer kind %s\n", kind.c_str()); return 5; }
    }
    auto dims = rd("net_in_dims.u64");
    ciphertext3D t = to_tensor(rd("net_in.u64"), dims[0], dims[1], dims[2]);
    ofstream dg(DIR + "/ref_digests.txt");
    // Network::forward (network.cpp:22-47).  The committed reference refreshes in front of layer 6 (network.cpp:23); here the layer comes from reenc.u64 (absent:
    // no refresh) and the refresh is the reference's OWN decryptImage / encryptImage (globals.cpp:207-230, 144-157) under the keys of sk.u64 / pk.u64.  The floats
    // the client sees go to ref_reenc_floats.u64 (one float's bits per word) and the time to ref_reenc_us.u64; digests after the refresh depend on the
    // re-encryption's randomness, the decrypted outputs do not
    auto reenc = rd("reenc.u64", false);
    const int layer_before_reenc = reenc.empty() ? -1 : (int)reenc[0];
    if (layer_before_reenc >= 0) {
        PublicKey pk = load_pk(rd("pk.u64")); SecretKey sk2 = load_sk(rd("sk.u64"));
        encryptor = new Encryptor(*context, pk); decryptor = new Decryptor(*context, sk2);
    }
    for (int i = 0; i < net.getNumLayers(); i++) {
        if (i == layer_before_reenc) {
            auto r0 = chrono::high_resolution_clock::now();
            floatCube image = decryptImage(t);
            t = encryptImage(image);
            auto r1 = chrono::high_resolution_clock::now();
            vector<u64> fl;
            for (auto &a : image) for (auto &b : a) for (float v : b) { uint32_t bits; memcpy(&bits, &v, 4); fl.push_back(bits); }
            wr("ref_reenc_floats.u64", fl);
            wr("ref_reenc_us.u64", vector<u64>{(u64)chrono::duration_cast<chrono::microseconds>(r1 - r0).count()});
            fprintf(stderr, "refresh in front of layer %d done\n", i);
        }
        auto t0 = chrono::high_resolution_clock::now();
        string lname;
        if (sliced.count(i)) {
            const Sl &sl = sliced[i]; lname = sl.name;
            auto w = rdf(sl.name + ".weight.f64"), b = rdf(sl.name + ".bias.f64");
            ciphertext3D res(1, ciphertext2D(sl.out, vector<Ciphertext>(1)));
            int per = (sl.out + sl.slices - 1) / sl.slices;
            for (int r0 = 0; r0 < sl.out; r0 += per) {
                int r1 = min(sl.out, r0 + per);
                vector<double> ws(w.begin() + (size_t)r0 * sl.in, w.begin() + (size_t)r1 * sl.in), bs(b.begin() + r0, b.begin() + r1);
                unique_ptr<FullyConnectedLayer> l(make_fc(sl.name, sl.in, r1 - r0, min(sl.th, r1 - r0), ws, bs));
                ciphertext3D part = l->forward(t);
                for (int r = r0; r < r1; r++) res[0][r][0] = part[0][r - r0][0];
                fprintf(stderr, "  rows %d..%d done\n", r0, r1);
            }
            t = res;
        } else { lname = net.getLayer(i)->getName(); t = net.getLayer(i)->forward(t); }
        auto t1 = chrono::high_resolution_clock::now();
        string d = digest(t);
        dg << i << " " << lname << " " << t.size() << "x" << t[0].size() << "x" << t[0][0].size() << " " << d << " "
           << chrono::duration_cast<chrono::microseconds>(t1 - t0).count() << "us" << endl;
        fprintf(stderr, "layer %d done\n", i);
    }
    vector<u64> out; from_tensor(t, out); wr("ref_net_out.u64", out);
    auto skv = rd("sk.u64", false);
    if (!skv.empty()) {      // client side of the reference: decrypt + decode the logits, report the remaining noise budget
        SecretKey sk = load_sk(skv); Decryptor dec(*context, sk);
        vector<u64> pl, logits, bud;
        for (auto &a : t) for (auto &b : a) for (auto &c : b) {
            Plaintext p; dec.decrypt(c, p); from_plain(p, pl, N);
            double v = fraencoder->decode(p); u64 bits; memcpy(&bits, &v, 8); logits.push_back(bits);
            bud.push_back(dec.invariant_noise_budget(c));
        }
        wr("ref_net_dec.u64", pl); wr("ref_net_logits.u64", logits); wr("ref_net_budget.u64", bud);
    }
    return 0;
}

int main(int argc, char **argv)
{
    if (argc < 3) { fprintf(stderr, "usage: ref_harness ops|layers|net <dir>\n"); return 1; }
    DIR = argv[2];
    string mode = argv[1];
    try {
        if (mode == "ops") return do_ops();
        if (mode == "layers") return do_layers();
        if (mode == "net") return do_net();
        if (mode == "files") return do_files();
    } catch (const exception &e) { fprintf(stderr, "reference threw: %s\n", e.what()); return 6; }
    return 1;
}
