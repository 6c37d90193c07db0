// crcnn_host.cpp -- implementation of the CrCNN-compatible C++ host classes on top of the C ABI (include/crcnn_hip.h).
#include "crcnn_host.h"
#include "../csrc/host_parallel.h"          // std::thread ranges (header only; no other csrc internals are used here: the engine is reached through the C ABI)
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iostream>
#include <list>
#include <mutex>

using namespace std;

crc_ctx *context = nullptr;
// The stream every ABI call of these classes is launched on.  NULL (the default stream) unless the caller installs its own (setStream): a driver that uploads
// the next chunk of encrypted images on a second stream while this one computes (bench_host's streamed inputs) needs a non-blocking one here -- the default
// stream would order itself against every other blocking stream of the process.  A global like the context, the keys and the work buffer
// (CrCNN/src/globals.h:18-26).
static void *g_stream = nullptr;
void setStream(void *s) { g_stream = s; }
void *getStream() { return g_stream; }
static inline void *stream() { return g_stream; }
vector<uint64_t> secret_key, public_key, ev_keys16_host;
shared_ptr<DeviceBuffer> ev_keys16;
static bool g_det = false;                                  // setDeterministicSeed(): tests / bench only
static uint64_t g_det_seed = 0;
static uint8_t g_master_key[CRC_KEY_BYTES];                 // fresh from the OS on every setParameters()
static uint64_t g_enc_counter = 0;                          // ciphertexts encrypted under g_master_key so far (= next keystream id)
void setDeterministicSeed(uint64_t seed) { g_det = true; g_det_seed = seed; }
void clearDeterministicSeed() { g_det = false; }

// ---- status -> exception (the reference throws std::invalid_argument from SEAL, evaluator.cpp:1549-1556) ---------------
static void chk(int status, const char *what)
{
    if (status >= 0) return;
    string msg = string(what) + ": " + crc_strerror(status);
    if (status == CRC_ERR_INVALID_ARGUMENT || status == CRC_ERR_PARAMETERS) throw invalid_argument(msg);
    throw runtime_error(msg);
}
static crc_ctx *ctx()
{
    if (!context) throw logic_error("setParameters() must be called first");
    return context;
}
static int N() { return crc_ctx_n(ctx()); }
static int K() { return crc_ctx_k(ctx()); }
static size_t ctBytes() { return crc_ct_words(ctx(), 2) * 8; }

// Device allocations are recycled by exact size: the reference passes its tensors by value and so do these classes -- every Layer::forward makes a new output
// tensor -- but a hipMalloc / hipFree pair per layer and chunk (the free synchronises the device) would cost more than some of the layers.  Freed buffers wait
// in a pool (bounded: beyond the cap they go back to the driver; an allocation that fails empties the pool and tries again); delParameters() empties it.
namespace {
struct BufferPool {
    std::mutex mu;
    std::list<std::pair<size_t, void *>> lru;               // oldest first
    size_t pooled = 0;
    static constexpr size_t kCap = (size_t)128 << 30, kReserve = (size_t)16 << 30;
    void *take(size_t b)
    {
        std::lock_guard<std::mutex> g(mu);
        for (auto it = lru.begin(); it != lru.end(); ++it) if (it->first == b) { void *p = it->second; lru.erase(it); pooled -= b; return p; }
        return nullptr;
    }
    // the newest buffer stays; the least recently returned ones (the one-off scratch of model building, typically) make room for it
    bool give(size_t b, void *p)
    {
        std::lock_guard<std::mutex> g(mu);
        if (b > kCap) return false;
        while (pooled + b > kCap && !lru.empty()) { if (context) crc_free(context, lru.front().second); pooled -= lru.front().first; lru.pop_front(); }
        // ... and the device keeps kReserve free for allocations that do not come through here (the HIP runtime's scratch for spilling kernels, RCCL)
        size_t free_b = 0, total_b = 0;
        while (context && crc_mem_info(context, &free_b, &total_b) >= 0 && free_b < kReserve && !lru.empty()) { crc_free(context, lru.front().second);
            pooled -= lru.front().first; lru.pop_front(); }
        if (context && free_b < kReserve) return false;
        lru.emplace_back(b, p); pooled += b;
        return true;
    }
    void flush() { std::lock_guard<std::mutex> g(mu); for (auto &e : lru) if (context) crc_free(context, e.second); lru.clear(); pooled = 0; }
};
BufferPool g_pool;
}
DeviceBuffer::DeviceBuffer(size_t b) : bytes(b)
{
    const size_t want = b ? b : 8;
    if ((ptr = g_pool.take(want))) return;
    static const bool trace = getenv("CRC_HOST_TRACE") != nullptr;
    if (trace) fprintf(stderr, "[host] hipMalloc %.3f GiB (pool miss)\n", want / 1073741824.0);
    if (crc_malloc(ctx(), want, &ptr) >= 0) return;
    g_pool.flush();
    chk(crc_malloc(ctx(), want, &ptr), "crc_malloc");
}
DeviceBuffer::~DeviceBuffer()
{
    if (!ptr || !context) return;
    if (g_pool.give(bytes ? bytes : 8, ptr)) return;
    if (getenv("CRC_HOST_TRACE")) fprintf(stderr, "[host] hipFree %.3f GiB (pool full)\n", bytes / 1073741824.0);
    crc_free(context, ptr);
}

// ---- Plaintext ------------------------------------------------------------------------------------------------------
void Plaintext::dense(uint64_t *out, int n) const
{
    memset(out, 0, 8 * (size_t)n);
    for (auto &p : nz) if (p.first < n) out[p.first] = p.second;
}
void Plaintext::save(ostream &stream) const
{
    int32_t cc = coeff_count_;
    vector<uint64_t> d((size_t)max(cc, 0), 0);
    for (auto &p : nz) if (p.first < cc) d[p.first] = p.second;
    stream.write(reinterpret_cast<const char *>(&cc), sizeof cc);
    stream.write(reinterpret_cast<const char *>(d.data()), (streamsize)d.size() * 8);
}
void Plaintext::load(istream &stream)
{
    int32_t cc = 0;
    stream.read(reinterpret_cast<char *>(&cc), sizeof cc);
    if (!stream || cc < 0 || cc > N() + 1) throw invalid_argument("plain is not valid for encryption parameters");
    vector<uint64_t> d((size_t)cc);
    stream.read(reinterpret_cast<char *>(d.data()), (streamsize)cc * 8);
    if (!stream) throw invalid_argument("truncated plaintext stream");
    coeff_count_ = cc; nz.clear();
    for (int i = 0; i < cc; i++) if (d[i]) nz.emplace_back(i, d[i]);
}
static Plaintext fromDense(const uint64_t *co, int n, int cc)
{
    Plaintext p; p.coeff_count_ = cc;
    for (int i = 0; i < n; i++) if (co[i]) p.nz.emplace_back(i, co[i]);
    return p;
}
Plaintext fraencode(double value)
{
    vector<uint64_t> co((size_t)N()); int32_t cc = 0;
    chk(crc_encode_f64(ctx(), &value, 1, co.data(), &cc), "crc_encode_f64");
    return fromDense(co.data(), N(), cc);
}
double fradecode(const vector<uint64_t> &plain) { return crc_decode(ctx(), plain.data()); }

// plaintext list -> device buffer of [count][k][n]: mode 0 = NTT-form weights, 1 = delta coefficient form, 2 = delta NTT form; mode 3 = the plaintext
// coefficients themselves, [count][n] (streamed layers)
static shared_ptr<DeviceBuffer> uploadPlain(const vector<const Plaintext *> &pl, int mode)
{
    const int n = N(), k = mode == 3 ? 1 : K();
    auto out = make_shared<DeviceBuffer>(pl.size() * (size_t)k * n * 8);
    const size_t chunk = max<size_t>(1, min<size_t>(pl.size(), (64u << 20) / ((size_t)n * 8)));
    DeviceBuffer stage(chunk * (size_t)n * 8), cstage(chunk * (size_t)CRC_PLAIN_COMPACT_WORDS * 8);
    vector<uint64_t> host, chost(chunk * (size_t)CRC_PLAIN_COMPACT_WORDS);
    for (size_t o = 0; o < pl.size(); o += chunk) {
        const size_t c = min(chunk, pl.size() - o);
        // what the fractional encoder produces has non-zero coefficients only at 0..63 and n-32..n-1: such plaintexts travel in compact form (96 words each,
        // crc_plain_expand zero-extends them on the device); anything else -- a plaintext loaded from a file may be arbitrary -- goes down dense
        atomic<bool> compact{true};
        crc_host::parallel_for(c, 64, [&](size_t b, size_t e) {
            for (size_t i = b; i < e; i++) {
                uint64_t *row = chost.data() + i * CRC_PLAIN_COMPACT_WORDS;
                memset(row, 0, sizeof(uint64_t) * CRC_PLAIN_COMPACT_WORDS);
                for (auto &z : pl[o + i]->nz) {
                    if (z.first < CRC_PLAIN_COMPACT_LOW) row[z.first] = z.second;
                    else if (z.first >= n - CRC_PLAIN_COMPACT_HIGH && z.first < n) row[CRC_PLAIN_COMPACT_LOW + z.first - (n - CRC_PLAIN_COMPACT_HIGH)] =
                        z.second;
                    else { compact = false; return; }
                }
            }
        });
        if (compact) {
            chk(crc_memcpy_h2d(ctx(), cstage.ptr, chost.data(), c * (size_t)CRC_PLAIN_COMPACT_WORDS * 8, stream()), "crc_memcpy_h2d");
            chk(crc_plain_expand(ctx(), (const uint64_t *)cstage.ptr, c, (uint64_t *)stage.ptr, stream()), "crc_plain_expand");
        } else {
            host.resize(chunk * (size_t)n);
            crc_host::parallel_for(c, 64, [&](size_t b, size_t e) { for (size_t i = b; i < e; i++) pl[o + i]->dense(host.data() + i * n, n); });
            chk(crc_memcpy_h2d(ctx(), stage.ptr, host.data(), c * (size_t)n * 8, stream()), "crc_memcpy_h2d");
        }
        uint64_t *dst = (uint64_t *)out->ptr + o * (size_t)k * n;
        if (mode == 3) chk(crc_memcpy_d2d(ctx(), dst, stage.ptr, c * (size_t)n * 8, stream()), "crc_memcpy_d2d");
        else if (mode == 0) chk(crc_plain_to_ntt(ctx(), (const uint64_t *)stage.ptr, c, dst, stream()), "crc_plain_to_ntt");
        else chk(crc_plain_to_delta(ctx(), (const uint64_t *)stage.ptr, c, mode == 2 ? CRC_NTT : CRC_COEFF, dst, stream()), "crc_plain_to_delta");
        chk(crc_stream_sync(ctx(), stream()), "crc_stream_sync");
    }
    return out;
}

// ---- tensors ---------------------------------------------------------------------------------------------------------- Network::forward's two ping-pong
// activation slots: the NEXT tensor constructed on this thread takes (and, if it is too small, replaces) the slot the hint points at instead of a buffer of its
// own -- every layer constructs its output tensor first.  With 200 GiB of weights resident there is no room for a recycling pool, and a hipMalloc / hipFree
// pair per layer costs more than most layers (PlainModelWoPad at n = 16384: 113 ms instead of 3.8 ms per image for conv1).
static thread_local shared_ptr<DeviceBuffer> *g_out_hint = nullptr;
// arms the hint for ONE layer call and disarms it when the scope ends, however it ends: a layer that throws before it has constructed its output must not leave
// the hint pointing at the network's slot for whatever tensor the caller constructs next (an encryptImage in a retry, or a slot of a Network that no longer
// exists)
struct OutHint {
    explicit OutHint(shared_ptr<DeviceBuffer> *slot) { g_out_hint = slot; }
    ~OutHint() { g_out_hint = nullptr; }
    OutHint(const OutHint &) = delete; OutHint &operator=(const OutHint &) = delete;
};

ciphertext3D::ciphertext3D(int B, int zd, int xd, int yd, int form) : B(B), zd(zd), xd(xd), yd(yd), form(form)
{
    size_t bytes = count() * ctBytes();
    if (form == CRC_NTTLC) { const size_t lb = crc_limb_tensor_bytes(ctx(), B, zd, xd, yd); if (lb > bytes) bytes = lb; }      // channels padded to 32
    // a dense consumer's limb tensor: every output a channel of ONE position, rounded up to 32 (7 bytes per residue: larger than the ciphertexts below 217
    // channels)
    if (form == CRC_NTTL) { const size_t lb = crc_limb_tensor_bytes(ctx(), B, zd * xd * yd, 1, 1); if (lb > bytes) bytes = lb; }
    if (g_out_hint) {
        shared_ptr<DeviceBuffer> *slot = g_out_hint; g_out_hint = nullptr;
        if (!*slot || (*slot)->bytes < bytes) { slot->reset(); *slot = make_shared<DeviceBuffer>(bytes); }
        buf = *slot;
        return;
    }
    buf = make_shared<DeviceBuffer>(bytes);
}
ciphertext3D ciphertext3D::images(int b0, int count) const
{
    if (!buf || b0 < 0 || count < 1 || b0 + count > B) throw invalid_argument("ciphertext3D::images: range outside the batch");
    if (form == CRC_NTTL || form == CRC_NTTLC) throw invalid_argument("ciphertext3D::images: a limb tensor is laid out for its whole batch");
    ciphertext3D v; v.B = count; v.zd = zd; v.xd = xd; v.yd = yd; v.form = form; v.buf = buf;
    v.offset = offset + (size_t)b0 * zd * xd * yd * ctBytes();
    return v;
}
ciphertext3D ciphertext3D::fromHost(const uint64_t *h, int B, int zd, int xd, int yd)
{
    ciphertext3D t(B, zd, xd, yd);
    chk(crc_memcpy_h2d(ctx(), t.buf->ptr, h, t.count() * ctBytes(), stream()), "crc_memcpy_h2d");
    chk(crc_stream_sync(ctx(), stream()), "crc_stream_sync");
    return t;
}
vector<uint64_t> ciphertext3D::toHost() const
{
    vector<uint64_t> h(count() * ctBytes() / 8);
    chk(crc_memcpy_d2h(ctx(), h.data(), data(), h.size() * 8, stream()), "crc_memcpy_d2h");
    chk(crc_stream_sync(ctx(), stream()), "crc_stream_sync");
    return h;
}
ciphertext3D stackImages(const vector<ciphertext3D> &images)
{
    if (images.empty()) throw invalid_argument("no images");
    const ciphertext3D &f = images[0];
    int B = 0;
    for (auto &im : images) {
        if (im.zd != f.zd || im.xd != f.xd || im.yd != f.yd || im.form != f.form) throw invalid_argument("image shapes differ");
        B += im.B;
    }
    ciphertext3D t(B, f.zd, f.xd, f.yd, f.form);
    size_t off = 0;
    for (auto &im : images) { chk(crc_memcpy_d2d(ctx(), (char *)t.data() + off, im.data(), im.count() * ctBytes(), stream()), "crc_memcpy_d2d");
        off += im.count() * ctBytes(); }
    chk(crc_stream_sync(ctx(), stream()), "crc_stream_sync");
    return t;
}
ciphertext3D deepCopyImage(const ciphertext3D &image)
{
    ciphertext3D t(image.B, image.zd, image.xd, image.yd, image.form);
    chk(crc_memcpy_d2d(ctx(), t.data(), image.data(), image.count() * ctBytes(), stream()), "crc_memcpy_d2d");
    return t;
}

// ---- globals ----------------------------------------------------------------------------------------------------------
void setParameters(int poly_modulus, uint64_t plain_modulus)
{
    uint64_t q[16];
    int k = crc_default_coeff_modulus_128(poly_modulus, q, 16);            // parms->set_coeff_modulus(coeff_modulus_128(n)), globals.cpp:30
    if (k < 0) throw invalid_argument("no default coeff_modulus for this poly_modulus");
    setParameters(poly_modulus, vector<uint64_t>(q, q + k), plain_modulus, 0);
}
void setParameters(int poly_modulus, const vector<uint64_t> &coeff_modulus, uint64_t plain_modulus, int device)
{
    delParameters();
    chk(crc_ctx_create(poly_modulus, coeff_modulus.data(), (int)coeff_modulus.size(), plain_modulus, device, &context),
        "encryption parameters are not set correctly");
    const int n = N(), k = K();
    secret_key.assign((size_t)k * n, 0); public_key.assign((size_t)2 * k * n, 0);
    ev_keys16_host.assign(crc_evk_words(context, 16), 0);                  // keygen->generate_evaluation_keys(16, *ev_keys16), globals.cpp:54
    g_enc_counter = 0;
    if (g_det) {
        chk(crc_keygen(context, g_det_seed, secret_key.data(), public_key.data()), "crc_keygen");
        chk(crc_gen_evk(context, g_det_seed + 1, secret_key.data(), 16, ev_keys16_host.data()), "crc_gen_evk");
    } else {
        chk(crc_random_key(g_master_key), "crc_random_key");
        chk(crc_keygen_key(context, g_master_key, secret_key.data(), public_key.data()), "crc_keygen_key");
        chk(crc_gen_evk_key(context, g_master_key, secret_key.data(), 16, ev_keys16_host.data()), "crc_gen_evk_key");
    }
    ev_keys16 = make_shared<DeviceBuffer>(ev_keys16_host.size() * 8);
    chk(crc_memcpy_h2d(context, ev_keys16->ptr, ev_keys16_host.data(), ev_keys16_host.size() * 8, stream()), "crc_memcpy_h2d");
    chk(crc_stream_sync(context, stream()), "crc_stream_sync");
}
// One kernel scratch area for every layer of the process: layers run one after another on one stream, so their scratch never overlaps in
// time, and the largest request decides the size (a per-layer buffer summed to ~40 GiB at WoPad 16384 beside the 182 GiB limb weights).
static shared_ptr<DeviceBuffer> g_scratch;
// limb-form tile buffers of the streamed layers (one set serves every streamed layer: they run one after the other)
static shared_ptr<DeviceBuffer> g_wltile, g_xltile;
// device copies of the client's keys (secret key for the refresh, public key for every encryption); uploaded on first use, dropped with the parameters
// (the host vectors are public globals, as in the reference: a caller may assign them -- a fingerprint of the words decides whether the copy is current)
static shared_ptr<DeviceBuffer> g_d_sk, g_d_pk;
static uint64_t g_d_sk_fp = 0, g_d_pk_fp = 0;
static const uint64_t *deviceKey(shared_ptr<DeviceBuffer> &d, uint64_t &fp, const vector<uint64_t> &h, const char *what)
{
    if (h.empty()) throw logic_error(string("setParameters() or initFromKeys() must be called first (") + what + ")");
    uint64_t f = 0x9e3779b97f4a7c15ULL ^ h.size();
    for (uint64_t w : h) f = (f ^ w) * 0xff51afd7ed558ccdULL + (f >> 29);
    if (!d || d->bytes != h.size() * 8 || f != fp) {
        fp = f;
        d = make_shared<DeviceBuffer>(h.size() * 8);
        chk(crc_memcpy_h2d(ctx(), d->ptr, h.data(), h.size() * 8, stream()), "crc_memcpy_h2d");
        chk(crc_stream_sync(ctx(), stream()), "crc_stream_sync");           // the host vector may change after this call returns
    }
    return (const uint64_t *)d->ptr;
}

void delParameters()
{
    ev_keys16.reset();
    g_d_sk.reset(); g_d_pk.reset();
    g_scratch.reset(); g_wltile.reset(); g_xltile.reset();
    g_pool.flush();
    if (context) { crc_ctx_destroy(context); context = nullptr; }
}
static void writeFile(const string &path, const vector<uint8_t> &b) { ofstream f(path, ofstream::binary); if (!f) throw runtime_error("cannot write " + path);
    f.write((const char *)b.data(), (streamsize)b.size()); }
static vector<uint8_t> readFile(const string &path)
{
    ifstream f(path, ifstream::binary); if (!f) throw runtime_error("cannot open " + path);
    f.seekg(0, ios::end); size_t sz = (size_t)f.tellg(); f.seekg(0); vector<uint8_t> b(sz); f.read((char *)b.data(), (streamsize)sz); return b;
}
void setAndSaveParameters(string public_key_path, string secret_key_path, string evaluation_key_path, int poly_modulus, uint64_t plain_modulus)
{   // globals.cpp:58-74
    setParameters(poly_modulus, plain_modulus);
    size_t w = 0;
    vector<uint8_t> b(crc_seal_pk_bytes(context)); chk(crc_seal_pk_save(context, public_key.data(), b.data(), b.size(), &w), "crc_seal_pk_save");
        writeFile(public_key_path, b);
    b.assign(crc_seal_sk_bytes(context), 0); chk(crc_seal_sk_save(context, secret_key.data(), b.data(), b.size(), &w), "crc_seal_sk_save");
        writeFile(secret_key_path, b);
    b.assign(crc_seal_evk_bytes(context, 16), 0); chk(crc_seal_evk_save(context, ev_keys16_host.data(), 16, b.data(), b.size(), &w), "crc_seal_evk_save");
        writeFile(evaluation_key_path, b);
}
void initFromKeys(string public_key_path, string secret_key_path, string evaluation_key_path, int poly_modulus, uint64_t plain_modulus)
{   // globals.cpp:77-111
    setParameters(poly_modulus, plain_modulus);
    vector<uint8_t> b = readFile(public_key_path); chk(crc_seal_pk_load(context, b.data(), b.size(), public_key.data()),
        "public_key is not valid for encryption parameters");
    b = readFile(secret_key_path); chk(crc_seal_sk_load(context, b.data(), b.size(), secret_key.data()), "secret_key is not valid for encryption parameters");
    b = readFile(evaluation_key_path); int dbc = 0;
    chk(crc_seal_evk_load(context, b.data(), b.size(), ev_keys16_host.data(), &dbc), "evaluation_keys is not valid for encryption parameters");
    if (dbc != 16) throw invalid_argument("evaluation keys must have decomposition_bit_count 16");
    chk(crc_memcpy_h2d(context, ev_keys16->ptr, ev_keys16_host.data(), ev_keys16_host.size() * 8, stream()), "crc_memcpy_h2d");
    chk(crc_stream_sync(context, stream()), "crc_stream_sync");
    g_d_sk.reset(); g_d_pk.reset();
}
ciphertext3D encryptAndSaveImage(vector<float> image, int zd, int xd, int yd, string file_name)
{   // globals.cpp:174-190: the ciphertexts back to back in Ciphertext::save format
    ciphertext3D t = encryptImage(image, zd, xd, yd);
    vector<uint64_t> h = t.toHost();
    const size_t one = crc_seal_ct_bytes(ctx(), 2), ctw = crc_ct_words(ctx(), 2);
    vector<uint8_t> b(one * t.count()); size_t w = 0;
    for (size_t i = 0; i < t.count(); i++) chk(crc_seal_ct_save(ctx(), h.data() + i * ctw, 2, b.data() + i * one, one, &w), "crc_seal_ct_save");
    writeFile(file_name, b);
    return t;
}
ciphertext3D loadEncryptedImage(int zd, int xd, int yd, string file_name)
{   // globals.cpp:193-205
    vector<uint8_t> b = readFile(file_name);
    const size_t cnt = (size_t)zd * xd * yd, ctw = crc_ct_words(ctx(), 2);
    vector<uint64_t> h(cnt * ctw); size_t off = 0;
    for (size_t i = 0; i < cnt; i++) { int size = 0; size_t used = 0;
        chk(crc_seal_ct_load(ctx(), b.data() + off, b.size() - off, h.data() + i * ctw, 2, &size, &used), "encrypted is not valid for encryption parameters");
        if (size != 2) throw invalid_argument("expected size-2 ciphertexts");
        off += used; }
    return ciphertext3D::fromHost(h.data(), 1, zd, xd, yd);
}
// encode on the host, encrypt on the device (crc_encrypt_dev: Encryptor::encrypt, encryptor.cpp:71-134)
static ciphertext3D encryptPixels(const vector<float> &px, int zd, int xd, int yd)
{
    const int n = N();
    vector<uint64_t> pl(px.size() * n);
    chk(crc_encode_f32(ctx(), px.data(), px.size(), pl.data(), nullptr), "crc_encode_f32");
    DeviceBuffer d_pl(pl.size() * 8), d_work(crc_encrypt_dev_work_bytes(ctx(), px.size()));
    const uint64_t *d_pk = deviceKey(g_d_pk, g_d_pk_fp, public_key, "public key");
    chk(crc_memcpy_h2d(ctx(), d_pl.ptr, pl.data(), pl.size() * 8, stream()), "crc_memcpy_h2d");
    ciphertext3D out(1, zd, xd, yd, CRC_COEFF);
    if (g_det)
        chk(crc_encrypt_dev(ctx(), d_pk, (const uint64_t *)d_pl.ptr, px.size(), g_det_seed + 1000003 * (g_enc_counter + 1),
                            (uint64_t *)out.buf->ptr, d_work.ptr, stream()), "crc_encrypt_dev");
    else                                                    // one keystream per ciphertext, never reused under this key
        chk(crc_encrypt_dev_key(ctx(), d_pk, (const uint64_t *)d_pl.ptr, px.size(), g_master_key, g_enc_counter,
                                (uint64_t *)out.buf->ptr, d_work.ptr, stream()), "crc_encrypt_dev_key");
    g_enc_counter += px.size();
    chk(crc_stream_sync(ctx(), stream()), "crc_stream_sync");
    return out;
}
ciphertext3D encryptImage(vector<float> image, int zd, int xd, int yd)
{
    if ((int)image.size() < xd * yd) throw invalid_argument("image too small");
    // the reference indexes image[i*xd+j] for every z (globals.cpp:133): one plane replicated over zd
    vector<float> px((size_t)zd * xd * yd);
    for (int z = 0; z < zd; z++) for (int i = 0; i < xd; i++) for (int j = 0; j < yd; j++) px[((size_t)z * xd + i) * yd + j] = image[(size_t)i * xd + j];
    return encryptPixels(px, zd, xd, yd);
}
ciphertext3D encryptImage(floatCube image)
{
    const int zd = (int)image.size(), xd = (int)image[0].size(), yd = (int)image[0][0].size();
    vector<float> px; px.reserve((size_t)zd * xd * yd);
    for (auto &a : image) for (auto &b : a) for (float v : b) px.push_back(v);
    return encryptPixels(px, zd, xd, yd);
}
vector<floatCube> decryptImages(const ciphertext3D &t)
{
    const int n = N();
    if (t.form != CRC_COEFF) throw invalid_argument("tensor is in NTT form");
    vector<uint64_t> h = t.toHost(), pl(t.count() * n);
    chk(crc_decrypt(ctx(), secret_key.data(), h.data(), t.count(), 2, pl.data()), "crc_decrypt");
    vector<floatCube> out(t.B, floatCube(t.zd, vector<vector<float>>(t.xd, vector<float>(t.yd))));
    size_t i = 0;
    for (int b = 0; b < t.B; b++) for (int z = 0; z < t.zd; z++) for (int x = 0; x < t.xd; x++) for (int y = 0; y < t.yd; y++, i++)
        out[b][z][x][y] = (float)crc_decode(ctx(), pl.data() + i * n);
    return out;
}
floatCube decryptImage(const ciphertext3D &t)
{
    if (t.B != 1) throw invalid_argument("decryptImage expects a single image; use decryptImages for a batch");
    return decryptImages(t)[0];
}
int noiseBudget(const ciphertext3D &t, size_t index)
{
    vector<uint64_t> h(ctBytes() / 8);
    chk(crc_memcpy_d2h(ctx(), h.data(), (char *)t.data() + index * ctBytes(), ctBytes(), stream()), "crc_memcpy_d2h");
    chk(crc_stream_sync(ctx(), stream()), "crc_stream_sync");
    return crc_noise_budget(ctx(), secret_key.data(), h.data(), 2);
}

// ---- Layer ------------------------------------------------------------------------------------------------------------
void Layer::computeBoundaries(int xd, int yd, int xs, int ys, int xf, int yf, int *xl, int *yl)
{
    *xl = xf > xs ? xd - xf + 1 : xd - xs + 1;
    *yl = yf > ys ? yd - yf + 1 : yd - ys + 1;
}
static void checkInput(const ciphertext3D &in, int zd, int xd, int yd, const char *who)
{
    if (!in.buf || in.zd != zd || in.xd != xd || in.yd != yd) throw invalid_argument(string(who) + ": input tensor shape does not match the layer");
}
static shared_ptr<DeviceBuffer> &ensure(shared_ptr<DeviceBuffer> &b, size_t bytes)
{
    if (!b || b->bytes < bytes) b = make_shared<DeviceBuffer>(bytes);
    return b;
}

// The client-side refresh of network.cpp:30-34 -- `floatCube image = decryptImage(input); input = encryptImage(image);` -- for a whole batch on the launch
// stream (crc_refresh_dev: decrypt, decode, round to float, encode, encrypt; nothing crosses PCIe and the host does not wait).  Passes of bounded size share the
// layers' scratch area.  Fresh randomness per ciphertext exactly as encryptImage draws it (deterministic only under setDeterministicSeed)
ciphertext3D refreshImages(const ciphertext3D &t, int out_form, vector<float> *values)
{
    if (!t.buf) throw invalid_argument("refreshImages: empty tensor");
    if ((t.form != CRC_COEFF && t.form != CRC_NTT) || (out_form != CRC_COEFF && out_form != CRC_NTT))
        throw invalid_argument("refreshImages: ciphertext forms only (CRC_COEFF / CRC_NTT)");
    const uint64_t *d_sk = deviceKey(g_d_sk, g_d_sk_fp, secret_key, "secret key"), *d_pk = deviceKey(g_d_pk, g_d_pk_fp, public_key, "public key");
    ciphertext3D out(t.B, t.zd, t.xd, t.yd, out_form);
    const size_t cnt = t.count(), one = crc_refresh_dev_work_bytes(ctx(), 1, t.form);
    size_t pass = ((size_t)4 << 30) / (one ? one : 1);
    if (pass < 1024) pass = 1024;
    if (pass > cnt) pass = cnt;
    ensure(g_scratch, crc_refresh_dev_work_bytes(ctx(), pass, t.form));
    shared_ptr<DeviceBuffer> d_vals;
    if (values) d_vals = make_shared<DeviceBuffer>(cnt * sizeof(float));
    for (size_t o = 0; o < cnt; o += pass) {
        const size_t c = min(pass, cnt - o);
        const uint64_t *in = (const uint64_t *)((const char *)t.data() + o * ctBytes());
        uint64_t *dst = (uint64_t *)((char *)out.data() + o * ctBytes());
        float *dv = d_vals ? (float *)d_vals->ptr + o : nullptr;
        if (g_det)
            chk(crc_refresh_dev(ctx(), d_sk, d_pk, in, c, t.form, g_det_seed + 1000003 * (g_enc_counter + 1), out_form, dst, dv, g_scratch->ptr, stream()),
                "crc_refresh_dev");
        else
            chk(crc_refresh_dev_key(ctx(), d_sk, d_pk, in, c, t.form, g_master_key, g_enc_counter, out_form, dst, dv, g_scratch->ptr, stream()),
                "crc_refresh_dev_key");
        g_enc_counter += c;
    }
    if (values) {
        values->assign(cnt, 0.f);
        chk(crc_memcpy_d2h(ctx(), values->data(), d_vals->ptr, cnt * sizeof(float), stream()), "crc_memcpy_d2h");
        chk(crc_stream_sync(ctx(), stream()), "crc_stream_sync");
    }
    return out;
}

static int g_expected_batch = 0;                            // images per Network::forward the caller announced (0: unknown)
void setExpectedBatch(int images_per_forward) { g_expected_batch = images_per_forward > 0 ? images_per_forward : 0; }
static bool tooLargeForHbm(size_t weights)
{
    size_t free_b = 0, total_b = 0;
    chk(crc_mem_info(ctx(), &free_b, &total_b), "crc_mem_info");
    const char *e = getenv("CRC_STREAM_SHARE");                 // (tests force streaming on small rings with a tiny share, as netrun.py does)
    const double share = e ? atof(e) : 0.75;
    return (double)weights * K() * N() * 8 > share * (double)total_b;
}
// a streamed layer: lift + NTT a tile of filters, run the layer on the tile, scatter the tile's output channels into the [B][F][P] tensor
static int plannedForm(int zd, int xd, int yd, int xs, int ys, int xf, int yf, int nf, int B);
static int forwardStreamed(const ciphertext3D &input, ciphertext3D &out, int zd, int xd, int yd, int xs, int ys, int xf, int yf, int nf, int out_form,
                            shared_ptr<DeviceBuffer> &d_plain, shared_ptr<DeviceBuffer> *d_b, shared_ptr<DeviceBuffer> &d_wtile,
                                shared_ptr<DeviceBuffer> &d_ytile, shared_ptr<DeviceBuffer> &d_work)
{
    const size_t n = N(), k = K(), rowb = k * n * 8, ctb = ctBytes();
    const size_t T = (size_t)zd * xf * yf, P = (size_t)((xd - xf) / xs + 1) * ((yd - yf) / ys + 1);
    auto scatter = [&](int f0, int ft) {
        for (int b = 0; b < input.B; b++)
            chk(crc_memcpy_d2d(ctx(), (char *)out.buf->ptr + ((size_t)b * nf + f0) * P * ctb, (const char *)d_ytile->ptr + (size_t)b * ft * P * ctb,
                (size_t)ft * P * ctb, stream()), "crc_memcpy_d2d");
    };
    // On the matrix cores (a reduction the limb GEMM takes, at least 32 rows in this launch: crc_plan_mac): tiles of 64 filters in limb form, built from
    // canonical sub-tiles of 8 filters (crc_limb_pack_weights_tile), the layer's input converted to limb form once per launch -- PlainModelWoPad's fc3 with all
    // eight primes of n = 16384 (netrun.py does the same: 25 against 58 ms per image on the vector-ALU tiles)
    const bool ntt_in = input.form == CRC_NTT || input.form == CRC_NTTP || input.form == CRC_NTTL;
    if (ntt_in && plannedForm(zd, xd, yd, xs, ys, xf, yf, nf, input.B) == CRC_NTTL) {
        const int ft_max = min(64, nf), sub = min(8, ft_max);
        const size_t wt = (size_t)sub * T * rowb, wl = crc_limb_weights_bytes(ctx(), ft_max, zd, xf, yf), yt = (size_t)input.B * ft_max * P * ctb;
        if (!d_wtile || d_wtile->bytes < wt) d_wtile = make_shared<DeviceBuffer>(wt);
        if (!d_ytile || d_ytile->bytes < yt) d_ytile = make_shared<DeviceBuffer>(yt);
        if (!g_wltile || g_wltile->bytes < wl) g_wltile = make_shared<DeviceBuffer>(wl);
        const void *xl = input.data();
        if (input.form != CRC_NTTL) {
            const size_t xb = crc_limb_tensor_bytes(ctx(), input.B, zd, xd, yd);
            if (!g_xltile || g_xltile->bytes < xb) g_xltile = make_shared<DeviceBuffer>(xb);
            chk(crc_limb_pack_tensor(ctx(), input.data(), input.form, input.B, zd, xd, yd, g_xltile->ptr, stream()), "crc_limb_pack_tensor");
            xl = g_xltile->ptr;
        }
        const size_t wb = crc_conv2d_forms_work_bytes(ctx(), input.B, zd, xd, yd, xs, ys, xf, yf, ft_max, CRC_NTTL, CRC_NTTL, out_form);
        if (!d_work || d_work->bytes < wb) d_work = make_shared<DeviceBuffer>(wb);
        for (int f0 = 0; f0 < nf; f0 += ft_max) {
            const int ft = min(ft_max, nf - f0);
            for (int s0 = 0; s0 < ft; s0 += sub) {
                const int fs = min(sub, ft - s0);
                chk(crc_plain_to_ntt(ctx(), (const uint64_t *)d_plain->ptr + (size_t)(f0 + s0) * T * n, (size_t)fs * T, (uint64_t *)d_wtile->ptr, stream()),
                    "crc_plain_to_ntt");
                chk(crc_limb_pack_weights_tile(ctx(), (const uint64_t *)d_wtile->ptr, ft, s0, fs, zd, xf, yf, g_wltile->ptr, stream()),
                    "crc_limb_pack_weights_tile");
            }
            chk(crc_conv2d_forms(ctx(), (const uint64_t *)xl, (const uint64_t *)g_wltile->ptr, CRC_NTTL,
                (const uint64_t *)((const char *)d_b[out_form != CRC_COEFF]->ptr + (size_t)f0 * rowb),
                                 input.B, zd, xd, yd, xs, ys, xf, yf, ft, CRC_NTTL, out_form, (uint64_t *)d_ytile->ptr, d_work->ptr, stream()),
                                     "crc_conv2d_forms");
            scatter(f0, ft);
        }
        return CRC_NTTL;
    }
    // tile: as many filters (a multiple of 8, the MAC kernel's filter granule) as make 2-16 GiB of NTT-form weights, by what HBM has left
    size_t free_b = 0, total_b = 0;
    chk(crc_mem_info(ctx(), &free_b, &total_b), "crc_mem_info");
    const size_t tile_bytes = d_wtile && d_wtile->bytes >= ((size_t)1 << 30) ? d_wtile->bytes : max<size_t>((size_t)2 << 30, min<size_t>((size_t)16 << 30,
        free_b / 8));
    size_t ftv = tile_bytes / (T * rowb); if (ftv >= 8) ftv = ftv / 8 * 8;
    const int ft_max = (int)max<size_t>(1, min<size_t>(nf, ftv));
    if (!d_wtile || d_wtile->bytes < ft_max * T * rowb) d_wtile = make_shared<DeviceBuffer>(ft_max * T * rowb);
    if (!d_ytile || d_ytile->bytes < (size_t)input.B * ft_max * P * ctb) d_ytile = make_shared<DeviceBuffer>((size_t)input.B * ft_max * P * ctb);
    size_t wb = crc_conv2d_forms_work_bytes(ctx(), input.B, zd, xd, yd, xs, ys, xf, yf, ft_max, input.form, CRC_NTT, out_form);
    if (!d_work || d_work->bytes < wb) d_work = make_shared<DeviceBuffer>(wb);
    for (int f0 = 0; f0 < nf; f0 += ft_max) {
        const int ft = min(ft_max, nf - f0);
        chk(crc_plain_to_ntt(ctx(), (const uint64_t *)d_plain->ptr + (size_t)f0 * T * n, (size_t)ft * T, (uint64_t *)d_wtile->ptr, stream()),
            "crc_plain_to_ntt");
        chk(crc_conv2d_forms(ctx(), input.data(), (const uint64_t *)d_wtile->ptr, CRC_NTT, (const uint64_t *)((const char *)d_b[out_form != CRC_COEFF]->ptr +
            (size_t)f0 * rowb), input.B,
                             zd, xd, yd, xs, ys, xf, yf, ft, input.form, out_form, (uint64_t *)d_ytile->ptr, d_work->ptr, stream()), "crc_conv2d_forms");
        scatter(f0, ft);
    }
    return CRC_NTT;
}

// the kernel crc_plan_mac picks for a conv / dense layer launched on B images (the one statement of the policy, shared with netrun.py)
static bool g_matrix_cores = true;                  // Network::matrix_cores of the forward in progress (layers called directly plan with the default)
static int plannedForm(int zd, int xd, int yd, int xs, int ys, int xf, int yf, int nf, int B)
{
    int wf = CRC_NTT;
    chk(crc_plan_mac(ctx(), zd, xd, yd, xs, ys, xf, yf, nf, B, g_matrix_cores ? 1 : 0, &wf), "crc_plan_mac");
    return wf;
}
// canonical NTT-form weights -> limb form (CRC_NTTL) when crc_plan_mac says the limb GEMM pays for this shape and launch size and the second copy fits beside
// the first; the canonical copy is dropped
static bool limbFits(int nf, int zd, int xf, int yf)
{
    size_t free_b = 0, total_b = 0;
    chk(crc_mem_info(ctx(), &free_b, &total_b), "crc_mem_info");
    return free_b >= crc_limb_weights_bytes(ctx(), nf, zd, xf, yf) + ((size_t)24 << 30);
}
static bool toLimb(shared_ptr<DeviceBuffer> &d_w, int &w_form, int nf, int zd, int xf, int yf, bool planned)
{
    if (w_form == CRC_NTTL) return true;
    if (!planned || !limbFits(nf, zd, xf, yf)) return false;
    const size_t nbytes = crc_limb_weights_bytes(ctx(), nf, zd, xf, yf);
    auto wl = make_shared<DeviceBuffer>(nbytes);
    chk(crc_limb_pack_weights(ctx(), (const uint64_t *)d_w->ptr, nf, zd, xf, yf, wl->ptr, stream()), "crc_limb_pack_weights");
    chk(crc_stream_sync(ctx(), stream()), "crc_stream_sync");
    d_w = wl; w_form = CRC_NTTL;
    return true;
}

// ---- ConvolutionalLayer -----------------------------------------------------------------------------------------------
ConvolutionalLayer::ConvolutionalLayer(string name, int xd, int yd, int zd, int xs, int ys, int xf, int yf, int nf, int th_count, plaintext4D &filters,
    vector<Plaintext> &biases)
    : Layer(name), xd(xd), yd(yd), zd(zd), xs(xs), ys(ys), xf(xf), yf(yf), nf(nf), th_count(th_count),
      xo((xd - xf) / xs + 1), yo((yd - yf) / ys + 1), zo(nf), filters(filters), biases(biases) {}
ConvolutionalLayer::ConvolutionalLayer(string name, int xd, int yd, int zd, int xs, int ys, int xf, int yf, int nf, int th_count, istream *infile)
    : Layer(name), xd(xd), yd(yd), zd(zd), xs(xs), ys(ys), xf(xf), yf(yf), nf(nf), th_count(th_count),
      xo((xd - xf) / xs + 1), yo((yd - yf) / ys + 1), zo(nf) { loadPlaintextParameters(infile); }
void ConvolutionalLayer::upload()
{
    if (filters_already_ntt) return;
    if ((int)filters.size() != nf || (int)biases.size() != nf) throw invalid_argument("conv: filter/bias count mismatch");
    vector<const Plaintext *> w, b;
    for (int f = 0; f < nf; f++) {
        if ((int)filters[f].size() != zd || (int)filters[f][0].size() != xf ||
            (int)filters[f][0][0].size() != yf) throw invalid_argument("conv: kernel shape mismatch");
        for (int z = 0; z < zd; z++) for (int i = 0; i < xf; i++) for (int j = 0; j < yf; j++) w.push_back(&filters[f][z][i][j]);
        b.push_back(&biases[f]);
    }
    streamed = forced_placement >= 0 ? forced_placement == 1 : tooLargeForHbm(w.size());
    if (streamed) d_plain = uploadPlain(w, 3); else d_w = uploadPlain(w, 0);
    d_b[0] = uploadPlain(b, 1); d_b[1] = uploadPlain(b, 2);
    filters_already_ntt = true;            // transform_kernel_to_ntt, convolutionalLayer.cpp:151-156 (done once)
}
static size_t bytesOf(const shared_ptr<DeviceBuffer> &b) { return b ? b->bytes : 0; }
static string macKernelName(int w_form, bool streamed)
{
    const string k = w_form == CRC_NTTL ? "mfma_mac2w_kernel (int8 limb GEMM, CRC_NTTL)" : w_form == CRC_NTTL1 ?
        "mfma_conv1_kernel (one-channel convolution on the matrix cores, CRC_NTTL1)"
                   : w_form == CRC_NTTP ? "mac3_kernel (v_mad_u64_u32, CRC_NTTP)" : "mac3_kernel (v_mad_u64_u32, canonical residues)";
    return streamed ? k + (w_form == CRC_NTTL ? ", streamed weights (64-filter limb tiles built inside the forward)" : ", streamed weights") : k;
}
size_t ConvolutionalLayer::deviceBytes() const { return bytesOf(d_w) + bytesOf(d_b[0]) + bytesOf(d_b[1]) + bytesOf(d_plain) + bytesOf(d_wtile) +
    bytesOf(d_ytile) + bytesOf(d_w_canon); }
string ConvolutionalLayer::kernelName() const { return macKernelName(streamed ? stream_form : w_form, streamed); }
int ConvolutionalLayer::placement() { upload(); return streamed ? 1 : 0; }
void ConvolutionalLayer::restoreCanonical()
{
    if (w_form == CRC_NTTP) { packWeights(true); return; }
    if (w_form == CRC_NTTL1 && d_w_canon) { d_w = d_w_canon; d_w_canon.reset(); w_form = CRC_NTT; return; }
    if (w_form != CRC_NTTL && w_form != CRC_NTTL1) return;
    if ((int)filters.size() != nf) throw logic_error("ConvolutionalLayer " + name +
        ": a folded layer's weights are in limb form and it has no plaintexts to rebuild them from");
    d_w.reset(); w_form = CRC_NTT; filters_already_ntt = false;
    upload();
}
void ConvolutionalLayer::deviceParameters(vector<shared_ptr<DeviceBuffer>> &out, bool allocate_only)
{
    if (allocate_only && !filters_already_ntt) {
        const size_t rowb = (size_t)K() * N() * 8;
        streamed = forced_placement >= 0 ? forced_placement == 1 : tooLargeForHbm((size_t)nf * zd * xf * yf);
        if (!streamed) d_w = make_shared<DeviceBuffer>((size_t)nf * zd * xf * yf * rowb);
        d_b[0] = make_shared<DeviceBuffer>(nf * rowb); d_b[1] = make_shared<DeviceBuffer>(nf * rowb);
        filters_already_ntt = true;
    }
    packWeights(true);                                      // canonical residues on the wire (uploads first if needed)
    if (streamed && !d_plain) d_plain = make_shared<DeviceBuffer>((size_t)nf * zd * xf * yf * N() * 8);
    out.push_back(streamed ? d_plain : d_w); out.push_back(d_b[0]); out.push_back(d_b[1]);
}
bool ConvolutionalLayer::limbWeights(int B)
{
    upload();
    if (streamed) return false;
    if (w_form == CRC_NTTL || w_form == CRC_NTTL1) return true;
    const int planned = plannedForm(zd, xd, yd, xs, ys, xf, yf, nf, B);
    // (decided BEFORE the weights are touched: a layer that stays on the vector-ALU kernel keeps its 28-bit packed weights -- unpacking and re-packing them on
    // every forward() is a read-modify-write of the whole layer)
    if (planned != CRC_NTTL1 && !(planned == CRC_NTTL && limbFits(nf, zd, xf, yf))) return false;
    if (w_form == CRC_NTTP) packWeights(true);
    if (planned == CRC_NTTL1) {          // one-channel convolutions have their own matrix-core kernel (kernels_mfma1.hip)
        auto wl = make_shared<DeviceBuffer>(crc_limb_conv1_weights_bytes(ctx()));
        chk(crc_limb_conv1_pack_weights(ctx(), (const uint64_t *)d_w->ptr, nf, xf, yf, wl->ptr, stream()), "crc_limb_conv1_pack_weights");
        chk(crc_stream_sync(ctx(), stream()), "crc_stream_sync");
        // (the canonical copy of a one-channel layer is small: kept, so that the weights can go back on the wire)
        d_w_canon = d_w; d_w = wl; w_form = CRC_NTTL1;
        return true;
    }
    return toLimb(d_w, w_form, nf, zd, xf, yf, planned == CRC_NTTL);
}
void ConvolutionalLayer::packWeights(bool unpack)
{
    upload();
    if (streamed) return;
    if (w_form == CRC_NTTL1 && unpack) { d_w = d_w_canon; d_w_canon.reset(); w_form = CRC_NTT; return; }
    if (w_form == CRC_NTTL || w_form == CRC_NTTL1) { if (unpack) throw logic_error("ConvolutionalLayer " + name +
        ": weights are in limb form (fuse() / broadcastParameters() must precede the first forward())"); return; }
    if ((w_form == CRC_NTTP) == !unpack) return;
    chk(crc_pack28(ctx(), (uint64_t *)d_w->ptr, (size_t)nf * zd * xf * yf * K(), unpack ? 1 : 0, stream()), "crc_pack28");
    w_form = unpack ? CRC_NTT : CRC_NTTP;
}
ciphertext3D ConvolutionalLayer::forward(ciphertext3D input)
{
    checkInput(input, zd, xd, yd, "ConvolutionalLayer");
    upload();
    ciphertext3D out(input.B, zo, xo, yo, out_form);
    if (streamed) { stream_form = forwardStreamed(input, out, zd, xd, yd, xs, ys, xf, yf, nf, out_form, d_plain, d_b, d_wtile, d_ytile, g_scratch);
        return out; }
    size_t wb = crc_conv2d_forms_work_bytes(ctx(), input.B, zd, xd, yd, xs, ys, xf, yf, nf, input.form, w_form, out_form);
    if (!wb) throw invalid_argument("ConvolutionalLayer: unsupported geometry");
    ensure(g_scratch, wb);
    chk(crc_conv2d_forms(ctx(), input.data(), (const uint64_t *)d_w->ptr, w_form, (const uint64_t *)d_b[out_form != CRC_COEFF]->ptr, input.B, zd, xd, yd, xs,
        ys, xf, yf, nf,
                         input.form, out_form, out.data(), g_scratch->ptr, stream()), "crc_conv2d_forms");
    if (out_form == CRC_NTTLC) out.form = CRC_NTTL;         // what the convolution behind reads as its limb-form input
    return out;
}
void ConvolutionalLayer::savePlaintextParameters(ostream *outfile)
{   // order of convolutionalLayer.cpp:213-229
    if ((int)filters.size() != nf) throw logic_error("ConvolutionalLayer " + name +
        ": a folded layer has no plaintext parameters to save (save before Network::fuse())");
    for (int n = 0; n < nf; n++) { for (int z = 0; z < zd; z++) for (int i = 0; i < xf; i++) for (int j = 0; j < yf; j++) filters[n][z][i][j].save(*outfile);
        biases[n].save(*outfile); outfile->flush(); }
}
void ConvolutionalLayer::loadPlaintextParameters(istream *infile)
{
    filters.assign(nf, plaintext3D(zd, plaintext2D(xf, vector<Plaintext>(yf)))); biases.assign(nf, Plaintext());
    for (int n = 0; n < nf; n++) { for (int z = 0; z < zd; z++) for (int i = 0; i < xf; i++) for (int j = 0; j < yf; j++) filters[n][z][i][j].load(*infile);
        biases[n].load(*infile); }
    filters_already_ntt = false;
}
void ConvolutionalLayer::printLayerStructure()
{
    cerr << "Convolutional " << name << " : input (" << zd << "," << xd << "," << yd << "); kernel(" << nf << "," << xf << "," << yf << "); stride(" << xs <<
        "," << ys << "); output("
         << zo << "," << xo << "," << yo << ") " << "run with " << th_count << " threads" << endl;
}

// ---- FullyConnectedLayer ----------------------------------------------------------------------------------------------
FullyConnectedLayer::FullyConnectedLayer(string name, int in_dim, int out_dim, int th_count, plaintext2D &weights, vector<Plaintext> &biases)
    : Layer(name), in_dim(in_dim), out_dim(out_dim), th_count(th_count), weights(weights), biases(biases) {}
FullyConnectedLayer::FullyConnectedLayer(string name, int in_dim, int out_dim, int th_count, istream *infile)
    : Layer(name), in_dim(in_dim), out_dim(out_dim), th_count(th_count) { loadPlaintextParameters(infile); }
void FullyConnectedLayer::upload()
{
    if (weights_already_ntt) return;
    if ((int)weights.size() != out_dim || (int)biases.size() != out_dim) throw invalid_argument("fc: weight/bias count mismatch");
    vector<const Plaintext *> w, b;
    for (int i = 0; i < out_dim; i++) {
        if ((int)weights[i].size() != in_dim) throw invalid_argument("fc: row length mismatch");
        for (int j = 0; j < in_dim; j++) w.push_back(&weights[i][j]);
        b.push_back(&biases[i]);
    }
    streamed = forced_placement >= 0 ? forced_placement == 1 : tooLargeForHbm(w.size());
    if (forced_placement >= 0) tilewise = forced_placement == 2;
    // (g_expected_batch: a deployment that evaluates one image at a time -- setExpectedBatch(1) -- never takes the limb GEMM for a dense layer, so its canonical
    // weights stay resident and the layer runs as a weight stream (mac_stream_kernel) instead of being built tile-wise in limb form)
    else if (!streamed && plannedForm(in_dim, 1, 1, 1, 1, 1, 1, out_dim, g_expected_batch) == CRC_NTTL) {
        // canonical + limb copy beyond what HBM has left, the limb copy alone within it: build the limb weights tile by tile at the first forward
        // (buildTilewise)
        size_t free_b = 0, total_b = 0;
        chk(crc_mem_info(ctx(), &free_b, &total_b), "crc_mem_info");
        const size_t canon = w.size() * (size_t)K() * N() * 8, limb = crc_limb_weights_bytes(ctx(), out_dim, in_dim, 1, 1), reserve = (size_t)24 << 30;
        // (the tests force it on small rings)
        tilewise = (canon + limb + reserve > free_b && limb + reserve + ((size_t)8 << 30) <= free_b) || getenv("CRC_FORCE_TILEWISE") != nullptr;
    }
    if (streamed) d_plain = uploadPlain(w, 3); else if (!tilewise) d_w = uploadPlain(w, 0);
    d_b[0] = uploadPlain(b, 1); d_b[1] = uploadPlain(b, 2);
    weights_already_ntt = true;
}
void FullyConnectedLayer::buildTilewise()
{
    if (tile_built) return;
    const int n = N(), k = K();
    const size_t rowb = (size_t)k * n * 8, T = (size_t)in_dim;
    d_w = make_shared<DeviceBuffer>(crc_limb_weights_bytes(ctx(), out_dim, in_dim, 1, 1));
    const int ft = (int)max<size_t>(1, min<size_t>((size_t)out_dim, ((size_t)4 << 30) / (T * rowb)));
    shared_ptr<DeviceBuffer> fake, outc, wk;
    vector<uint64_t> bias, corr, q(k);
    int ch = 0, per_ch = 0;
    if (fold_bn) {
        ch = fold_bn->num_channels; per_ch = in_dim / ch;
        fake = make_shared<DeviceBuffer>(T * 2 * rowb); outc = make_shared<DeviceBuffer>((size_t)ft * 2 * rowb);
        wk = make_shared<DeviceBuffer>(max<size_t>(crc_dense_work_bytes(ctx(), 1, in_dim, ft, CRC_NTT), 256));
        chk(crc_memset(ctx(), fake->ptr, 0, T * 2 * rowb, stream()), "crc_memset");
        for (int z = 0; z < ch; z++) for (int t = 0; t < per_ch; t++)
            chk(crc_memcpy_d2d(ctx(), (char *)fake->ptr + ((size_t)z * per_ch + t) * 2 * rowb, (char *)fold_bn->d_mean[1]->ptr + (size_t)z * rowb, rowb,
                stream()), "crc_memcpy_d2d");
        bias.resize((size_t)out_dim * k * n); corr.resize((size_t)ft * 2 * k * n);
        chk(crc_memcpy_d2h(ctx(), bias.data(), d_b[1]->ptr, bias.size() * 8, stream()), "crc_memcpy_d2h");
        chk(crc_stream_sync(ctx(), stream()), "crc_stream_sync");
        chk(crc_ctx_table(ctx(), "q", q.data(), k) < 0 ? CRC_ERR_INVALID_ARGUMENT : CRC_OK, "crc_ctx_table");
    }
    for (int f0 = 0; f0 < out_dim; f0 += ft) {
        const int fn = min(ft, out_dim - f0);
        vector<const Plaintext *> w;
        for (int i = f0; i < f0 + fn; i++) for (int j = 0; j < in_dim; j++) w.push_back(&weights[i][j]);
        shared_ptr<DeviceBuffer> wt = uploadPlain(w, 0);                                   // lift + NTT of the tile's plaintexts (canonical, scratch)
        if (fold_bn) {
            for (int f = 0; f < fn; f++)                                                   // w'[f][z][tap] = w (*) s[z]
                chk(crc_multiply_plain_ntt(ctx(), (uint64_t *)wt->ptr + (size_t)f * T * k * n, (const uint64_t *)fold_bn->d_invstd->ptr, T, per_ch, 1,
                    stream()), "crc_multiply_plain_ntt");
            chk(crc_dense(ctx(), (const uint64_t *)fake->ptr, (const uint64_t *)wt->ptr, nullptr, 1, in_dim, fn, CRC_NTT, CRC_NTT, (uint64_t *)outc->ptr,
                wk->ptr, stream()), "crc_dense");
            chk(crc_memcpy_d2h(ctx(), corr.data(), outc->ptr, (size_t)fn * 2 * rowb, stream()), "crc_memcpy_d2h");
            chk(crc_stream_sync(ctx(), stream()), "crc_stream_sync");
            for (int f = 0; f < fn; f++) for (int m = 0; m < k; m++) for (int s2 = 0; s2 < n; s2++) {
                uint64_t &b = bias[((size_t)(f0 + f) * k + m) * n + s2]; const uint64_t c = corr[(((size_t)f * 2) * k + m) * n + s2];
                b = b >= c ? b - c : b + q[m] - c;
            }
        }
        chk(crc_limb_pack_weights_tile(ctx(), (const uint64_t *)wt->ptr, out_dim, f0, fn, in_dim, 1, 1, d_w->ptr, stream()), "crc_limb_pack_weights_tile");
        chk(crc_stream_sync(ctx(), stream()), "crc_stream_sync");
    }
    if (fold_bn) {
        d_b[1] = make_shared<DeviceBuffer>(bias.size() * 8);
        chk(crc_memcpy_h2d(ctx(), d_b[1]->ptr, bias.data(), bias.size() * 8, stream()), "crc_memcpy_h2d");
        chk(crc_stream_sync(ctx(), stream()), "crc_stream_sync");
        d_b[0] = make_shared<DeviceBuffer>(bias.size() * 8);
        chk(crc_memcpy_d2d(ctx(), d_b[0]->ptr, d_b[1]->ptr, bias.size() * 8, stream()), "crc_memcpy_d2d");
        chk(crc_ntt_inv(ctx(), (uint64_t *)d_b[0]->ptr, (size_t)out_dim, 1, stream()), "crc_ntt_inv");
        chk(crc_stream_sync(ctx(), stream()), "crc_stream_sync");
    }
    w_form = CRC_NTTL; tile_built = true;
}
size_t FullyConnectedLayer::deviceBytes() const { return bytesOf(d_w) + bytesOf(d_b[0]) + bytesOf(d_b[1]) + bytesOf(d_plain) + bytesOf(d_wtile) +
    bytesOf(d_ytile); }
string FullyConnectedLayer::kernelName() const { if (last_B == 1 && !streamed && (w_form == CRC_NTTP || w_form == CRC_NTT)) return w_form == CRC_NTTP ?
    "mac_stream_kernel (weight stream: one image, two rows per weight; v_mad_u64_u32, CRC_NTTP)" : "mac_stream_kernel (weight stream: one image; canonical residues)";
    return macKernelName(streamed ? stream_form : w_form, streamed) + (tilewise ?
    ", limb weights built tile by tile" : ""); }
int FullyConnectedLayer::placement() { upload(); return streamed ? 1 : tilewise ? 2 : 0; }
bool FullyConnectedLayer::streamsOnMatrixCores(int B) { upload(); return streamed && plannedForm(in_dim, 1, 1, 1, 1, 1, 1, out_dim, B) == CRC_NTTL; }
void FullyConnectedLayer::restoreCanonical()
{
    if (w_form == CRC_NTTP) { packWeights(true); return; }
    if (w_form != CRC_NTTL) return;
    if ((int)weights.size() != out_dim) throw logic_error("FullyConnectedLayer " + name +
        ": weights are in limb form and there are no plaintexts to rebuild them from");
    // (a tile-wise layer goes back to "not built": the next forward builds its limb tensor again, with whatever batch-norm layer fuse() folds into it)
    d_w.reset(); w_form = CRC_NTT; weights_already_ntt = false; tile_built = false;
    upload();
}
void FullyConnectedLayer::deviceParameters(vector<shared_ptr<DeviceBuffer>> &out, bool allocate_only)
{
    if (allocate_only && !weights_already_ntt && forced_placement == 2) {
        // a tile-wise layer is never on the wire (its only device copy is the limb tensor -- 182 GiB for PlainModelWoPad's fc3 at n = 16384, k = 4 -- which
        // every rank builds from its own plaintexts, deterministically): a receiving rank needs the model's plaintexts like the root
        if ((int)weights.size() != out_dim) throw logic_error("FullyConnectedLayer " + name +
            ": tile-wise weights are built on every rank -- a receiving rank must load the model too");
    } else if (allocate_only && !weights_already_ntt) {
        const size_t rowb = (size_t)K() * N() * 8;
        streamed = forced_placement >= 0 ? forced_placement == 1 : tooLargeForHbm((size_t)in_dim * out_dim);
        if (!streamed) d_w = make_shared<DeviceBuffer>((size_t)in_dim * out_dim * rowb);
        d_b[0] = make_shared<DeviceBuffer>(out_dim * rowb); d_b[1] = make_shared<DeviceBuffer>(out_dim * rowb);
        weights_already_ntt = true;
    }
    upload();
    if (tilewise) { buildTilewise(); return; }              // nothing to send or receive (see above)
    packWeights(true);
    if (streamed && !d_plain) d_plain = make_shared<DeviceBuffer>((size_t)in_dim * out_dim * N() * 8);
    out.push_back(streamed ? d_plain : d_w); out.push_back(d_b[0]); out.push_back(d_b[1]);
}
bool FullyConnectedLayer::limbWeights(int B)
{
    upload();
    if (streamed) return false;
    if (tilewise) { buildTilewise(); return true; }
    if (w_form == CRC_NTTL) return true;
    // (before the packed weights are touched)
    if (plannedForm(in_dim, 1, 1, 1, 1, 1, 1, out_dim, B) != CRC_NTTL || !limbFits(out_dim, in_dim, 1, 1)) return false;
    if (w_form == CRC_NTTP) packWeights(true);
    return toLimb(d_w, w_form, out_dim, in_dim, 1, 1, true);
}
void FullyConnectedLayer::packWeights(bool unpack)
{
    upload();
    if (streamed || (tilewise && !tile_built)) return;
    if (w_form == CRC_NTTL) { if (unpack) throw logic_error("FullyConnectedLayer " + name +
        ": weights are in limb form (fuse() / broadcastParameters() must precede the first forward())"); return; }
    if ((w_form == CRC_NTTP) == !unpack) return;
    chk(crc_pack28(ctx(), (uint64_t *)d_w->ptr, (size_t)in_dim * out_dim * K(), unpack ? 1 : 0, stream()), "crc_pack28");
    w_form = unpack ? CRC_NTT : CRC_NTTP;
}
ciphertext3D FullyConnectedLayer::forward(ciphertext3D input)
{
    // reshapeInput, :38-56
    if (!input.buf || input.zd * input.xd * input.yd != in_dim) throw invalid_argument("FullyConnectedLayer: input size does not match in_dim");
    upload();
    // a tile-wise layer has no canonical weights: whoever reaches it first -- Network::forward through limbWeights, a direct call, a network with matrix_cores
    // off -- builds the limb tensor, the only form its weights exist in (the layer then runs on the limb GEMM whatever the plan would have been)
    if (tilewise && !tile_built) buildTilewise();
    last_B = input.B;
    ciphertext3D out(input.B, 1, out_dim, 1, out_form);
    if (streamed) { stream_form = forwardStreamed(input, out, in_dim, 1, 1, 1, 1, 1, 1, out_dim, out_form, d_plain, d_b, d_wtile, d_ytile, g_scratch);
        return out; }
    ensure(g_scratch, crc_conv2d_forms_work_bytes(ctx(), input.B, in_dim, 1, 1, 1, 1, 1, 1, out_dim, input.form, w_form, out_form));
    chk(crc_dense_forms(ctx(), input.data(), (const uint64_t *)d_w->ptr, w_form, (const uint64_t *)d_b[out_form != CRC_COEFF]->ptr, input.B, in_dim, out_dim,
        input.form, out_form,
                        out.data(), g_scratch->ptr, stream()), "crc_dense_forms");
    return out;
}
void FullyConnectedLayer::savePlaintextParameters(ostream *outfile)
{
    for (int i = 0; i < out_dim; i++) { for (int j = 0; j < in_dim; j++) weights[i][j].save(*outfile); biases[i].save(*outfile); outfile->flush(); }
}
void FullyConnectedLayer::loadPlaintextParameters(istream *infile)
{
    weights.assign(out_dim, vector<Plaintext>(in_dim)); biases.assign(out_dim, Plaintext());
    for (int i = 0; i < out_dim; i++) { for (int j = 0; j < in_dim; j++) weights[i][j].load(*infile); biases[i].load(*infile); }
    weights_already_ntt = false;
}
void FullyConnectedLayer::printLayerStructure() { cerr << "Fully connected " << name << " : (" << in_dim << " -> " << out_dim << ")" << "run with " <<
    th_count << " threads" << endl; }

// ---- Pooling ----------------------------------------------------------------------------------------------------------
PoolingLayer::PoolingLayer(string name, int xd, int yd, int zd, int xs, int ys, int xf, int yf)
    : Layer(name), xd(xd), yd(yd), zd(zd), xs(xs), ys(ys), xf(xf), yf(yf), xo((xd - xf) / xs + 1), yo((yd - yf) / ys + 1), zo(zd) {}
ciphertext3D PoolingLayer::forward(ciphertext3D input)
{
    checkInput(input, zd, xd, yd, "PoolingLayer");
    ciphertext3D out(input.B, zo, xo, yo, input.form);
    chk(crc_pool(ctx(), input.data(), input.B, zd, xd, yd, xs, ys, xf, yf, d_div ? (const uint64_t *)d_div->ptr : nullptr, input.form, out.data(), stream()),
        "crc_pool");
    if (out_form != out.form) {      // pooling is form-preserving; convert only if the network asked for the other form
        if (out_form == CRC_NTT) chk(crc_ntt_fwd(ctx(), out.data(), out.count(), 2, stream()), "crc_ntt_fwd");
            else chk(crc_ntt_inv(ctx(), out.data(), out.count(), 2, stream()), "crc_ntt_inv");
        out.form = out_form;
    }
    return out;
}
void PoolingLayer::printLayerStructure()
{
    cerr << "Pooling " << name << " : input (" << zo << "," << xd << "," << yd << "); kernel(" << xf << "," << yf << "); stride(" << xs << "," << ys <<
        "); output(" << zo << "," << xo << "," << yo << ")" << endl;
}
AvgPoolingLayer::AvgPoolingLayer(string name, int xd, int yd, int zd, int xs, int ys, int xf, int yf) : PoolingLayer(name, xd, yd, zd, xs, ys, xf, yf)
{
    div_factor = fraencode(1. / (xf * yf));                 // avgPoolingLayer.cpp:12
    d_div = uploadPlain({&div_factor}, 0);
}

// ---- Square -----------------------------------------------------------------------------------------------------------
ciphertext3D SquareLayer::forward(ciphertext3D input)
{
    if (!input.buf) throw invalid_argument("SquareLayer: empty input");
    if (!ev_keys16) throw invalid_argument("not enough evaluation keys");
    // either form in, the requested form out: crc_square_relin_forms keeps an NTT-resident network resident
    ciphertext3D out(input.B, input.zd, input.xd, input.yd, out_form);
    ensure(g_scratch, crc_square_relin_work_bytes(ctx(), input.count(), 16));
    chk(crc_square_relin_forms(ctx(), input.data(), input.form, input.count(), (const uint64_t *)ev_keys16->ptr, 16, out.data(), out_form, g_scratch->ptr,
        stream()),
        "crc_square_relin_forms");
    return out;
}
void SquareLayer::printLayerStructure() { cerr << "Square run with " << th_count << " threads" << endl; }

// ---- Square + pooling (Network::fuse) ---------------------------------------------------------------------------------
SquarePoolLayer::SquarePoolLayer(string name, int xd, int yd, int zd, int xs, int ys, int xf, int yf, int th_count, shared_ptr<DeviceBuffer> d_div)
    : Layer(name), xd(xd), yd(yd), zd(zd), xs(xs), ys(ys), xf(xf), yf(yf), xo((xd - xf) / xs + 1), yo((yd - yf) / ys + 1), zo(zd), th_count(th_count),
        d_div(d_div) {}
ciphertext3D SquarePoolLayer::forward(ciphertext3D input)
{
    checkInput(input, zd, xd, yd, "SquarePoolLayer");
    if (!ev_keys16) throw invalid_argument("not enough evaluation keys");
    // an average pooling's divisor multiplies slot-wise: the pooled tensor is made NTT-resident for it, and brought back to coefficients when the network asked
    // for those.  The packed / limb operand forms are not produced here (Network::forward never asks this layer for them)
    if (out_form != CRC_NTT && out_form != CRC_COEFF) throw invalid_argument("SquarePoolLayer: out_form must be CRC_NTT or CRC_COEFF");
    const int of = d_div ? CRC_NTT : out_form;
    ciphertext3D out(input.B, zo, xo, yo, of);
    ensure(g_scratch, crc_square_pool_relin_work_bytes(ctx(), input.B, zd, xd, yd, xs, ys, xf, yf, 16));
    chk(crc_square_pool_relin_forms(ctx(), input.data(), input.form, input.B, zd, xd, yd, xs, ys, xf, yf, (const uint64_t *)ev_keys16->ptr, 16,
                                    d_div ? (const uint64_t *)d_div->ptr : nullptr, out.data(), of, g_scratch->ptr, stream()), "crc_square_pool_relin_forms");
    if (of != out_form) { chk(crc_ntt_inv(ctx(), out.data(), out.count(), 2, stream()), "crc_ntt_inv"); out.form = out_form; }
    return out;
}
void SquarePoolLayer::printLayerStructure()
{
    cerr << "Square + Pooling " << name << " : input (" << zd << "," << xd << "," << yd << "); kernel(" << xf << "," << yf << "); stride(" << xs << "," <<
        ys << "); output(" << zo << "," << xo << ","
         << yo << "); one key switch per pooled ciphertext" << endl;
}

// ---- BatchNorm --------------------------------------------------------------------------------------------------------
BatchNormLayer::BatchNormLayer(string name, int num_channels, vector<Plaintext> &mean, vector<Plaintext> &var) : Layer(name), num_channels(num_channels),
    mean(mean), var(var) {}
BatchNormLayer::BatchNormLayer(string name, int num_channels, istream *infile) : Layer(name), num_channels(num_channels) { loadPlaintextParameters(infile); }
void BatchNormLayer::upload()
{
    if (d_invstd) return;
    if ((int)mean.size() != num_channels || (int)var.size() != num_channels) throw invalid_argument("bn: parameter count mismatch");
    vector<const Plaintext *> m, v;
    for (int i = 0; i < num_channels; i++) { m.push_back(&mean[i]); v.push_back(&var[i]); }
    d_mean[0] = uploadPlain(m, 1); d_mean[1] = uploadPlain(m, 2); d_invstd = uploadPlain(v, 0);
}
void BatchNormLayer::deviceParameters(vector<shared_ptr<DeviceBuffer>> &out, bool allocate_only)
{
    if (allocate_only && !d_invstd) {
        const size_t rowb = (size_t)K() * N() * 8;
        d_mean[0] = make_shared<DeviceBuffer>(num_channels * rowb); d_mean[1] = make_shared<DeviceBuffer>(num_channels * rowb);
            d_invstd = make_shared<DeviceBuffer>(num_channels * rowb);
    }
    upload();
    out.push_back(d_mean[0]); out.push_back(d_mean[1]); out.push_back(d_invstd);
}
ciphertext3D BatchNormLayer::forward(ciphertext3D input)
{
    if (!input.buf || input.zd != num_channels) throw invalid_argument("BatchNormLayer: channel count mismatch");
    upload();
    ciphertext3D out = deepCopyImage(input);                // the reference works on its by-value copy (batchNormLayer.cpp:29)
    chk(crc_batchnorm(ctx(), out.data(), out.B, out.zd, out.xd, out.yd, (const uint64_t *)d_mean[out.form == CRC_NTT]->ptr, (const uint64_t *)d_invstd->ptr,
        out.form, stream()), "crc_batchnorm");
    if (out_form != out.form) {
        if (out_form == CRC_NTT) chk(crc_ntt_fwd(ctx(), out.data(), out.count(), 2, stream()), "crc_ntt_fwd");
            else chk(crc_ntt_inv(ctx(), out.data(), out.count(), 2, stream()), "crc_ntt_inv");
        out.form = out_form;
    }
    return out;
}
void BatchNormLayer::savePlaintextParameters(ostream *outfile) { for (int i = 0; i < num_channels; i++) { mean[i].save(*outfile); var[i].save(*outfile);
    outfile->flush(); } }
void BatchNormLayer::loadPlaintextParameters(istream *infile)
{
    mean.assign(num_channels, Plaintext()); var.assign(num_channels, Plaintext());
    for (int i = 0; i < num_channels; i++) { mean[i].load(*infile); var[i].load(*infile); }
    d_invstd.reset();
}
void BatchNormLayer::printLayerStructure() { cerr << "BatchNormLayer2D " << name << " :num_channels " << num_channels << endl; }

// ---- Network ----------------------------------------------------------------------------------------------------------
void Network::printNetworkStructure()
{
    for (size_t i = 0; i < layers.size(); i++) { cerr << "(" << i << ") : "; layers[i]->printLayerStructure(); cout << endl; }
}
ciphertext3D Network::forward(ciphertext3D input)
{   // network.cpp:22-47
    const int L = (int)layers.size();
    // the planning flag is this forward's only: layers called directly afterwards plan with the default again
    struct Restore { bool &ref; bool old; ~Restore() { ref = old; } } restore_matrix_cores{g_matrix_cores, g_matrix_cores};
    g_matrix_cores = matrix_cores;
    // choose the form of every boundary: NTT between linear layers when resident, coefficient form into Square and out of the net
    // conv / dense weights go into the MAC kernels' operand form (28-bit limb pairs) once; moduli above 55 bits cannot be packed
    bool packable = true;
    { vector<uint64_t> q(K()); crc_ctx_table(ctx(), "q", q.data(), K()); for (uint64_t v : q) if (v >> 55) packable = false; }
    auto isMac = [&](int i) { return i >= 0 && i < L && (dynamic_pointer_cast<ConvolutionalLayer>(layers[i]) ||
        dynamic_pointer_cast<FullyConnectedLayer>(layers[i])); };
    vector<char> limb(L, 0), streams(L, 0);
    // two-level chunking: the layers in front of the first dense layer on sub-batches of head_chunk images, the dense layers on the whole batch
    int split = L;
    if (head_chunk > 0 && input.B > head_chunk && ntt_resident && max_num_of_reencryptions < 0)
        for (int i = 1; i < L; i++) if (dynamic_pointer_cast<FullyConnectedLayer>(layers[i])) { split = i; break; }
    const bool chunked = split < L;
    if (packable)
        for (int i = 0; i < L; i++) {
            const int Bi = chunked && i < split ? head_chunk : input.B;
            if (auto c = dynamic_pointer_cast<ConvolutionalLayer>(layers[i])) { limb[i] = matrix_cores && c->limbWeights(Bi);
                if (!limb[i]) c->packWeights(false); }
            else if (auto f = dynamic_pointer_cast<FullyConnectedLayer>(layers[i])) {
                limb[i] = matrix_cores && f->limbWeights(Bi);
                if (!limb[i]) f->packWeights(false);
                // a STREAMED dense layer that will run on the matrix cores (64-filter limb tiles built inside the forward) reads a limb tensor like a resident
                // one: the chunks of a group are packed straight into it, and no second copy of the group's input is made inside the layer
                if (matrix_cores && f->streamsOnMatrixCores(Bi)) { limb[i] = 1; streams[i] = 1; }
            }
        }
    for (int i = 0; i < L; i++) {
        // the tensor in front of the refresh is decrypted as it stands (crc_refresh_dev takes either ciphertext form): an NTT-resident network stays resident
        // across it, but no packed / limb hand-over spans it
        const bool before_refresh = i + 1 == layer_before_reenc;
        bool coeff = !ntt_resident || i == L - 1;
        // a conv / dense layer feeding another one hands its tensor over packed as well ... and a limb layer feeding a DENSE limb layer hands it over in limb
        // form (not across the chunk boundary: a dense layer's limb tensor is laid out for its whole batch, the chunks are assembled into it below)
        const bool to_dense_limb = i + 1 < L && limb[i] && !streams[i] && limb[i + 1] && dynamic_pointer_cast<FullyConnectedLayer>(layers[i + 1]) &&
            !(chunked && i + 1 == split);
        // ... and a one-channel convolution writes the limb tensor of a matrix-core CONVOLUTION behind it itself
        auto ci = dynamic_pointer_cast<ConvolutionalLayer>(layers[i]);
        auto cn = i + 1 < L ? dynamic_pointer_cast<ConvolutionalLayer>(layers[i + 1]) : nullptr;
        const bool to_conv_limb = ci && cn && ci->w_form == CRC_NTTL1 && cn->w_form == CRC_NTTL;
        layers[i]->out_form = coeff ? CRC_COEFF : before_refresh ? CRC_NTT : to_dense_limb && max_num_of_reencryptions < 0 ? CRC_NTTL : to_conv_limb &&
            max_num_of_reencryptions < 0 ? CRC_NTTLC : (packable && max_num_of_reencryptions < 0 && isMac(i) && isMac(i + 1) ? CRC_NTTP : CRC_NTT);
    }
    last_layer_ms.assign(L, 0.0);
    last_layer_launches.assign(L, 0);
    last_reenc_ms = 0.0;
    // one timed Layer::forward call: events on the launch stream (read after the last layer), or wall clock + stream synchronisation
    vector<pair<int, pair<void *, void *>>> timed;
    size_t ev_used = 0;
    if (time_with_events && !event_pool) event_pool = make_shared<EventPool>();
    auto next_event = [&]() { auto &ev = event_pool->ev; if (ev_used == ev.size()) { void *e = nullptr; chk(crc_event_create(ctx(), &e), "crc_event_create");
        ev.push_back(e); } return ev[ev_used++]; };
    auto run_layer = [&](int i, const ciphertext3D &in) {
        last_layer_launches[i]++;
        if (time_with_events) {
            void *e0 = next_event(), *e1 = next_event();
            chk(crc_event_record(ctx(), e0, stream()), "crc_event_record");
            ciphertext3D out = layers[i]->forward(in);
            chk(crc_event_record(ctx(), e1, stream()), "crc_event_record");
            timed.push_back({i, {e0, e1}});
            return out;
        }
        auto t0 = chrono::high_resolution_clock::now();
        ciphertext3D out = layers[i]->forward(in);
        chk(crc_stream_sync(ctx(), stream()), "crc_stream_sync");
        last_layer_ms[i] += chrono::duration<double, milli>(chrono::high_resolution_clock::now() - t0).count();
        return out;
    };
    auto read_events = [&]() {
        for (auto &t : timed) { float ms = 0; chk(crc_event_elapsed_ms(ctx(), t.second.first, t.second.second, &ms), "crc_event_elapsed_ms");
            (t.first < 0 ? last_reenc_ms : last_layer_ms[t.first]) += ms; }
        timed.clear();
    };
    // the refresh (network.cpp:30-34), timed like a layer: T_REENC of mainparams.cpp:81
    last_reenc_values.clear();
    auto refresh_into = [&](const ciphertext3D &in, int of) {
        if (!keep_reenc_values) return refreshImages(in, of);
        vector<float> v; ciphertext3D out = refreshImages(in, of, &v);
        last_reenc_values.insert(last_reenc_values.end(), v.begin(), v.end());
        return out;
    };
    auto run_refresh = [&](const ciphertext3D &in) {
        const int of = ntt_resident && max_num_of_reencryptions < 0 ? CRC_NTT : CRC_COEFF;
        OutHint hint(&act_slot[in.buf == act_slot[0] ? 1 : 0]);
        if (time_with_events) {
            void *e0 = next_event(), *e1 = next_event();
            chk(crc_event_record(ctx(), e0, stream()), "crc_event_record");
            ciphertext3D out = refresh_into(in, of);
            chk(crc_event_record(ctx(), e1, stream()), "crc_event_record");
            timed.push_back({-1, {e0, e1}});
            return out;
        }
        auto r0 = chrono::high_resolution_clock::now();
        ciphertext3D out = refresh_into(in, of);
        chk(crc_stream_sync(ctx(), stream()), "crc_stream_sync");
        last_reenc_ms += chrono::duration<double, milli>(chrono::high_resolution_clock::now() - r0).count();
        return out;
    };
    if (max_num_of_reencryptions >= 0) {                    // network.cpp:52-96
        int refreshes_left = max_num_of_reencryptions;
        for (int i = 0; i < L; i++) {
            layers[i]->out_form = CRC_COEFF;
            ciphertext3D output = run_layer(i, input);
            if (noiseBudget(output) <= 5) {
                if (refreshes_left <= 0) throw OutOfBudgetException(i - 1);
                input = run_refresh(input);
                refreshes_left--;
                i--;
                continue;
            }
            input = output;
        }
        read_events();
        return input;
    }
    int first = 0;
    if (chunked) {
        const int B = input.B;
        ciphertext3D tail_in;
        for (int b0 = 0; b0 < B; b0 += head_chunk) {
            const int Bc = min(head_chunk, B - b0);
            ciphertext3D t = input.images(b0, Bc);
            for (int i = 0; i < split; i++) {
                if (i == layer_before_reenc) t = run_refresh(t);
                OutHint hint(&act_slot[t.buf == act_slot[0] ? 1 : 0]); t = run_layer(i, t);
            }
            if (split == layer_before_reenc) t = run_refresh(t);        // in front of the first dense layer: chunk by chunk, before the chunks are assembled
            const size_t out_cts = (size_t)t.zd * t.xd * t.yd;
            if (!tail_in.buf) {         // kept across calls like the activation slots: next to 182 GiB of weights the pool has no room to hold it
                OutHint hint(&tail_slot);
                tail_in = ciphertext3D(B, t.zd, t.xd, t.yd, limb[split] ? CRC_NTTL : t.form);
            }
            if (limb[split])      // every chunk's tensor goes straight into the dense layer's K-blocked limb tensor
                chk(crc_limb_pack_tensor_at(ctx(), t.data(), t.form, Bc, (int)out_cts, 1, 1, tail_in.data(), B, b0, stream()), "crc_limb_pack_tensor_at");
            else
                chk(crc_memcpy_d2d(ctx(), (char *)tail_in.data() + (size_t)b0 * out_cts * ctBytes(), t.data(), (size_t)Bc * out_cts * ctBytes(), stream()),
                    "crc_memcpy_d2d");
        }
        chk(crc_stream_sync(ctx(), stream()), "crc_stream_sync");
        input = tail_in;
        first = split;
    }
    for (int i = first; i < L; i++) {
        // client-side refresh (needs the secret key), network.cpp:30-34; timed as T_REENC (:29-37).  Under two-level chunking a refresh at or in front of the
        // first dense layer has already run, chunk by chunk
        if (i == layer_before_reenc && !(chunked && layer_before_reenc <= split)) input = run_refresh(input);
        // every layer but the last writes into one of the network's two activation slots (the one its input does not live in); the last layer's output -- ten
        // ciphertexts per image -- is the caller's own tensor, as in the reference
        if (i + 1 < L) { OutHint hint(&act_slot[input.buf == act_slot[0] ? 1 : 0]); input = run_layer(i, input); }
        else {
            ciphertext3D output = run_layer(i, input);
            // a last layer that hands its input back (none of CrCNN's does) must not give the caller a tensor that lives in an activation slot the next forward
            // overwrites
            if (output.buf && (output.buf == act_slot[0] || output.buf == act_slot[1] || output.buf == tail_slot)) {
                ciphertext3D own(output.B, output.zd, output.xd, output.yd, output.form);
                chk(crc_memcpy_d2d(ctx(), own.data(), output.data(), output.count() * ctBytes(), stream()), "crc_memcpy_d2d");
                output = own;
            }
            input = output;
        }
    }
    if (time_with_events) { chk(crc_stream_sync(ctx(), stream()), "crc_stream_sync"); read_events(); }
    return input;
}
Network::HbmPlan Network::hbmPlan() const
{
    HbmPlan p;
    for (auto &l : layers) p.parameters += l->deviceBytes();
    p.activations = bytesOf(act_slot[0]) + bytesOf(act_slot[1]) + bytesOf(tail_slot);
    p.work = bytesOf(g_scratch);
    p.keys = bytesOf(ev_keys16);
    return p;
}
Network::EventPool::~EventPool() { if (context) for (void *e : ev) crc_event_destroy(context, e); }

size_t Network::broadcastParameters(crc_comm *comm, int root, bool encode_locally)
{
    if (!comm) throw invalid_argument("broadcastParameters: no communicator");
    const int rank = crc_comm_rank(comm), world = crc_comm_world(comm);
    if (root < 0 || root >= world) throw invalid_argument("broadcastParameters: bad root");
    // the root's placement of every layer's weights (resident / streamed / tile-wise) sizes the buffers on the wire: every rank adopts it before it allocates
    {
        const size_t L = layers.size();
        vector<uint64_t> mine_pl(L, 0), all_pl(L * (size_t)world, 0);
        if (rank == root) for (size_t i = 0; i < L; i++) mine_pl[i] = (uint64_t)layers[i]->placement();
        if (L > 64) throw invalid_argument("broadcastParameters: more than 64 layers");
        if (L) chk(crc_comm_allgather_u64(comm, mine_pl.data(), L, all_pl.data(), stream()), "crc_comm_allgather_u64");
        if (rank != root) for (size_t i = 0; i < L; i++) layers[i]->adoptPlacement((int)all_pl[(size_t)root * L + i]);
    }
    vector<shared_ptr<DeviceBuffer>> bufs;
    // encode_locally: every rank lifts + transforms its own plaintext parameters (it read the same model file) and only the evaluation keys travel; the
    // checksums below then prove that the ranks' encoders agree bit for bit
    for (auto &l : layers) l->deviceParameters(bufs, rank != root && !encode_locally);
    if (!ev_keys16) throw logic_error("setParameters() must be called first");
    const size_t own = encode_locally ? bufs.size() : 0;    // buffers that stay off the wire
    bufs.push_back(ev_keys16);                              // the evaluation keys come from the client through the root
    size_t bytes = 0;
    uint64_t mine[2] = {0, 0};
    for (size_t bi = 0; bi < bufs.size(); bi++) {
        auto &b = bufs[bi];
        const size_t words = b->bytes / 8;
        if (bi >= own) chk(crc_broadcast_weights(comm, (uint64_t *)b->ptr, words, root, stream()), "crc_broadcast_weights");
        uint64_t cs[2];
        chk(crc_checksum64(ctx(), (const uint64_t *)b->ptr, words, cs, stream()), "crc_checksum64");
        mine[0] ^= cs[0]; mine[1] = mine[1] * 0x9E3779B97F4A7C15ULL + cs[1];
        if (bi >= own) bytes += b->bytes;
    }
    vector<uint64_t> all((size_t)2 * world);
    chk(crc_comm_allgather_u64(comm, mine, 2, all.data(), stream()), "crc_comm_allgather_u64");
    for (int r = 0; r < world; r++)
        if (all[2 * r] != all[2 * root] || all[2 * r + 1] != all[2 * root + 1])
            throw runtime_error("broadcastParameters: rank " + to_string(r) + " holds different parameter bytes than the root");
    if (rank != root) {                                     // host copy of the keys follows the device copy
        chk(crc_memcpy_d2h(ctx(), ev_keys16_host.data(), ev_keys16->ptr, ev_keys16_host.size() * 8, stream()), "crc_memcpy_d2h");
        chk(crc_stream_sync(ctx(), stream()), "crc_stream_sync");
    }
    return bytes;
}

int Network::fuse()
{
    const int n = N(), k = K();
    const size_t rowb = (size_t)k * n * 8;
    int removed = 0;
    // The client-side refresh sits in front of layers[layer_before_reenc] (network.cpp:30-34).  No fold may span it, and the index follows the layers it
    // counts: erasing a layer below it moves the refresh point down with the layers behind it, so the refresh still runs in front of the same layer.
    auto refreshBetween = [&](size_t i) { return layer_before_reenc >= 0 && (int)i + 1 == layer_before_reenc; };
    auto eraseLayer = [&](size_t idx) {
        layers.erase(layers.begin() + idx);
        if (layer_before_reenc >= 0 && (int)idx < layer_before_reenc) layer_before_reenc--;
        removed++;
    };
    for (auto &l : layers) {                                // the folding kernels work on canonical residues
        // (a network that has already run holds its weights in the MAC kernels' operand forms: packed residues are unpacked, matrix-core forms -- which drop
        // the canonical copy -- are rebuilt from the layer's plaintexts, so fuse() may follow a forward())
        if (auto c = dynamic_pointer_cast<ConvolutionalLayer>(l)) c->restoreCanonical();
        else if (auto f = dynamic_pointer_cast<FullyConnectedLayer>(l)) f->restoreCanonical();
    }
    auto inttCopy = [&](const shared_ptr<DeviceBuffer> &ntt_rows, size_t rows) {       // coefficient-form twin of NTT-form delta rows
        auto out = make_shared<DeviceBuffer>(rows * rowb);
        chk(crc_memcpy_d2d(ctx(), out->ptr, ntt_rows->ptr, rows * rowb, stream()), "crc_memcpy_d2d");
        chk(crc_ntt_inv(ctx(), (uint64_t *)out->ptr, rows, 1, stream()), "crc_ntt_inv");
        return out;
    };
    // 1. conv + pool
    for (size_t i = 0; i + 1 < layers.size(); i++) {
        auto conv = dynamic_pointer_cast<ConvolutionalLayer>(layers[i]);
        auto pool = dynamic_pointer_cast<PoolingLayer>(layers[i + 1]);
        if (!conv || !pool || refreshBetween(i)) continue;
        conv->upload();
        if (conv->streamed) continue;
        if (pool->zd != conv->nf || pool->xd != conv->xo || pool->yd != conv->yo) continue;
        const int xf2 = (pool->xf - 1) * conv->xs + conv->xf, yf2 = (pool->yf - 1) * conv->ys + conv->yf;
        const int xs2 = conv->xs * pool->xs, ys2 = conv->ys * pool->ys;
        if (xf2 > conv->xd || yf2 > conv->yd) continue;
        const int xo2 = (conv->xd - xf2) / xs2 + 1, yo2 = (conv->yd - yf2) / ys2 + 1;
        if (xo2 != pool->xo || yo2 != pool->yo) continue;
        // the cost model lives behind the C ABI (crc_plan_fold_pool), shared with netrun.py
        const long long T2 = (long long)conv->zd * xf2 * yf2;
        int fold = 0;
        chk(crc_plan_fold_pool(ctx(), conv->zd, conv->xd, conv->yd, conv->xs, conv->ys, conv->xf, conv->yf, conv->nf, pool->xs, pool->ys, pool->xf, pool->yf,
            &fold), "crc_plan_fold_pool");
        if (!fold) continue;
        conv->upload();
        vector<Plaintext> nob; plaintext4D nof;
        auto fused = make_shared<ConvolutionalLayer>(conv->name + "+" + pool->name, conv->xd, conv->yd, conv->zd, xs2, ys2, xf2, yf2, conv->nf,
            conv->th_count, nof, nob);
        fused->d_w = make_shared<DeviceBuffer>((size_t)conv->nf * T2 * rowb);
        fused->d_b[1] = make_shared<DeviceBuffer>((size_t)conv->nf * rowb);
        chk(crc_conv2d_fold_pool(ctx(), (const uint64_t *)conv->d_w->ptr, (const uint64_t *)conv->d_b[1]->ptr, pool->d_div ?
            (const uint64_t *)pool->d_div->ptr : nullptr,
                                 conv->nf, conv->zd, conv->xf, conv->yf, conv->xs, conv->ys, pool->xf, pool->yf, (uint64_t *)fused->d_w->ptr,
                                     (uint64_t *)fused->d_b[1]->ptr, stream()),
            "crc_conv2d_fold_pool");
        fused->d_b[0] = inttCopy(fused->d_b[1], conv->nf);
        fused->filters_already_ntt = true;
        layers[i] = fused;
        eraseLayer(i + 1);
    }
    // 1b. Square + pooling: one key switch per pooled ciphertext (SquarePoolLayer)
    for (size_t i = 0; i + 1 < layers.size(); i++) {
        auto sq = dynamic_pointer_cast<SquareLayer>(layers[i]);
        auto pool = dynamic_pointer_cast<PoolingLayer>(layers[i + 1]);
        if (!sq || !pool || refreshBetween(i)) continue;
        if (!crc_square_pool_relin_supported(ctx(), 16, pool->xf, pool->yf)) continue;
        layers[i] = make_shared<SquarePoolLayer>(sq->name + "+" + pool->name, pool->xd, pool->yd, pool->zd, pool->xs, pool->ys, pool->xf, pool->yf,
            sq->th_count,
                                                 pool->d_div);
        eraseLayer(i + 1);
    }
    // 2. batch-norm + conv / dense
    for (size_t i = 0; i + 1 < layers.size(); i++) {
        auto bn = dynamic_pointer_cast<BatchNormLayer>(layers[i]);
        if (!bn || refreshBetween(i)) continue;
        auto conv = dynamic_pointer_cast<ConvolutionalLayer>(layers[i + 1]);
        auto fc = dynamic_pointer_cast<FullyConnectedLayer>(layers[i + 1]);
        if (!conv && !fc) continue;
        const int ch = bn->num_channels;
        int F, per_ch, T;
        if (conv) { if (conv->zd != ch) continue; F = conv->nf; per_ch = conv->xf * conv->yf; T = conv->zd * per_ch; conv->upload();
            if (conv->streamed) continue; }
        else { if (fc->in_dim % ch) continue; F = fc->out_dim; per_ch = fc->in_dim / ch; T = fc->in_dim; fc->upload(); if (fc->streamed) continue; }
        bn->upload();
        // no canonical weights to fold into: the fold is applied tile by tile when the limb weights are built
        if (fc && fc->tilewise) {
            if (fc->tile_built) continue;
            fc->fold_bn = bn;
            layers[i + 1]->name = bn->name + "+" + layers[i + 1]->name;
            eraseLayer(i);
            continue;
        }
        shared_ptr<DeviceBuffer> &dw = conv ? conv->d_w : fc->d_w;
        shared_ptr<DeviceBuffer> *db = conv ? conv->d_b : fc->d_b;
        // w'[f][z][tap] = w (*) s[z]
        for (int f = 0; f < F; f++)
            chk(crc_multiply_plain_ntt(ctx(), (uint64_t *)dw->ptr + (size_t)f * T * k * n, (const uint64_t *)bn->d_invstd->ptr, T, per_ch, 1, stream()),
                "crc_multiply_plain_ntt");
        // correction[f] = sum_t w'[f][t] (*) M[z(t)]: the dense kernel on one pseudo-image whose ciphertexts are (M[z(t)], 0)
        DeviceBuffer fake((size_t)T * 2 * rowb), outc((size_t)F * 2 * rowb), wk(max<size_t>(crc_dense_work_bytes(ctx(), 1, T, F, CRC_NTT), 256));
        chk(crc_memset(ctx(), fake.ptr, 0, (size_t)T * 2 * rowb, stream()), "crc_memset");
        for (int z = 0; z < ch; z++) for (int t = 0; t < per_ch; t++)
            chk(crc_memcpy_d2d(ctx(), (char *)fake.ptr + ((size_t)z * per_ch + t) * 2 * rowb, (char *)bn->d_mean[1]->ptr + (size_t)z * rowb, rowb, stream()),
                "crc_memcpy_d2d");
        chk(crc_dense(ctx(), (const uint64_t *)fake.ptr, (const uint64_t *)dw->ptr, nullptr, 1, T, F, CRC_NTT, CRC_NTT, (uint64_t *)outc.ptr, wk.ptr,
            stream()), "crc_dense");
        vector<uint64_t> corr((size_t)F * 2 * k * n), bias((size_t)F * k * n), q(k);
        chk(crc_memcpy_d2h(ctx(), corr.data(), outc.ptr, corr.size() * 8, stream()), "crc_memcpy_d2h");
        chk(crc_memcpy_d2h(ctx(), bias.data(), db[1]->ptr, bias.size() * 8, stream()), "crc_memcpy_d2h");
        chk(crc_stream_sync(ctx(), stream()), "crc_stream_sync");
        chk(crc_ctx_table(ctx(), "q", q.data(), k) < 0 ? CRC_ERR_INVALID_ARGUMENT : CRC_OK, "crc_ctx_table");
        for (int f = 0; f < F; f++) for (int m = 0; m < k; m++) for (int s2 = 0; s2 < n; s2++) {
            uint64_t &b = bias[((size_t)f * k + m) * n + s2]; const uint64_t c = corr[(((size_t)f * 2) * k + m) * n + s2];
            b = b >= c ? b - c : b + q[m] - c;
        }
        db[1] = make_shared<DeviceBuffer>(bias.size() * 8);
        chk(crc_memcpy_h2d(ctx(), db[1]->ptr, bias.data(), bias.size() * 8, stream()), "crc_memcpy_h2d");
        chk(crc_stream_sync(ctx(), stream()), "crc_stream_sync");
        db[0] = inttCopy(db[1], F);
        layers[i + 1]->name = bn->name + "+" + layers[i + 1]->name;
        eraseLayer(i);
    }
    chk(crc_stream_sync(ctx(), stream()), "crc_stream_sync");
    return removed;
}

// ---- CnnBuilder -------------------------------------------------------------------------------------------------------
vector<float> CnnBuilder::getPretrained(string var_name)
{   // LoadH5::getData, cnnBuilder.cpp:20-23
    size_t cnt = 0;
    int rc = crc_h5_dataset_count(plain_model_path.c_str(), var_name.c_str(), &cnt);
    if (rc) throw runtime_error("cannot read dataset " + var_name + " from " + plain_model_path + ": " + crc_strerror(rc));
    vector<float> v(cnt);
    chk(crc_h5_read_f32(plain_model_path.c_str(), var_name.c_str(), v.data(), cnt, nullptr), "crc_h5_read_f32");
    return v;
}
static vector<Plaintext> encodeAll(const vector<float> &v)
{
    // compact form (crc_encode_f32_compact: the 96 coefficients the encoder can set), encoded and turned into Plaintexts on the host threads
    // (csrc/host_parallel.h)
    const int n = N();
    vector<uint64_t> cp(v.size() * (size_t)CRC_PLAIN_COMPACT_WORDS); vector<int32_t> cc(v.size());
    chk(crc_encode_f32_compact(ctx(), v.data(), v.size(), cp.data(), cc.data()), "crc_encode_f32_compact");
    vector<Plaintext> out(v.size());
    crc_host::parallel_for(v.size(), 256, [&](size_t b, size_t e) {
        for (size_t i = b; i < e; i++) {
            const uint64_t *row = cp.data() + i * CRC_PLAIN_COMPACT_WORDS;
            Plaintext &p = out[i]; p.coeff_count_ = cc[i];
            for (int j = 0; j < CRC_PLAIN_COMPACT_LOW; j++) if (row[j]) p.nz.emplace_back(j, row[j]);
            for (int j = 0; j < CRC_PLAIN_COMPACT_HIGH; j++) if (row[CRC_PLAIN_COMPACT_LOW + j]) p.nz.emplace_back(n - CRC_PLAIN_COMPACT_HIGH + j,
                row[CRC_PLAIN_COMPACT_LOW + j]);
        }
    });
    return out;
}
ConvolutionalLayer *CnnBuilder::buildConvolutionalLayer(string name, int xd, int yd, int zd, int xs, int ys, int xf, int yf, int nf, int th_count,
    istream *infile)
{   // cnnBuilder.cpp:25-50
    if (infile != NULL) return new ConvolutionalLayer(name, xd, yd, zd, xs, ys, xf, yf, nf, th_count, infile);
    vector<float> weights = getPretrained(name + ".weight"), biases = getPretrained(name + ".bias");
    if ((int)weights.size() != nf * zd * xf * yf || (int)biases.size() != nf) throw invalid_argument("conv: dataset size does not match the layer");
    vector<Plaintext> ew = encodeAll(weights), eb = encodeAll(biases);
    plaintext4D encoded_weights(nf, plaintext3D(zd, plaintext2D(xf, vector<Plaintext>(yf))));
    size_t w = 0;
    for (int n = 0; n < nf; n++) for (int z = 0; z < zd; z++) for (int i = 0; i < xf; i++) for (int j = 0; j < yf; j++) encoded_weights[n][z][i][j] = ew[w++];
    return new ConvolutionalLayer(name, xd, yd, zd, xs, ys, xf, yf, nf, th_count, encoded_weights, eb);
}
FullyConnectedLayer *CnnBuilder::buildFullyConnectedLayer(string name, int in_dim, int out_dim, int th_count, istream *infile)
{   // cnnBuilder.cpp:53-76
    if (infile != NULL) return new FullyConnectedLayer(name, in_dim, out_dim, th_count, infile);
    vector<float> weights = getPretrained(name + ".weight"), biases = getPretrained(name + ".bias");
    if ((int)weights.size() != in_dim * out_dim || (int)biases.size() != out_dim) throw invalid_argument("fc: dataset size does not match the layer");
    vector<Plaintext> ew = encodeAll(weights), eb = encodeAll(biases);
    plaintext2D encoded_weights(out_dim, vector<Plaintext>(in_dim));
    size_t w = 0;
    for (int i = 0; i < out_dim; i++) for (int j = 0; j < in_dim; j++) encoded_weights[i][j] = ew[w++];
    return new FullyConnectedLayer(name, in_dim, out_dim, th_count, encoded_weights, eb);
}
PoolingLayer *CnnBuilder::buildPoolingLayer(string name, int xd, int yd, int zd, int xs, int ys, int xf, int yf) { return new PoolingLayer(name, xd, yd, zd,
    xs, ys, xf, yf); }
AvgPoolingLayer *CnnBuilder::buildAvgPoolingLayer(string name, int xd, int yd, int zd, int xs, int ys, int xf, int yf) { return new AvgPoolingLayer(name, xd,
    yd, zd, xs, ys, xf, yf); }
SquareLayer *CnnBuilder::buildSquareLayer(string name, int th_count) { return new SquareLayer(name, th_count); }
BatchNormLayer *CnnBuilder::buildBatchNormLayer(string name, int num_channels, istream *infile)
{   // cnnBuilder.cpp:89-105
    if (infile != NULL) return new BatchNormLayer(name, num_channels, infile);
    vector<float> mean = getPretrained(name + ".running_mean"), var = getPretrained(name + ".running_var");
    if ((int)mean.size() != num_channels || (int)var.size() != num_channels) throw invalid_argument("bn: dataset size does not match the layer");
    vector<float> invstd(var.size());
    chk(crc_bn_invstd_f32(var.data(), var.size(), invstd.data()), "crc_bn_invstd_f32");
    vector<Plaintext> em = encodeAll(mean), ev = encodeAll(invstd);
    return new BatchNormLayer(name, num_channels, em, ev);
}
Network CnnBuilder::buildNetwork(string file_name) { return buildNetworkByName("PlainModelTiny", file_name); }
Network CnnBuilder::buildNetworkByName(const string &model, string file_name)
{
    int th_count = 40, th_count2 = 50, th_tiny = 32, th_tiny2 = 42;
    Network net;
    unique_ptr<ifstream> infile;
    if (file_name != "") { infile.reset(new ifstream(file_name, ifstream::binary)); if (!*infile) throw runtime_error("cannot open " + file_name); }
    istream *in = infile.get();
    auto add = [&](Layer *l) { net.getLayers().push_back(shared_ptr<Layer>(l)); };
    if (model == "PlainModelTiny") {                         // cnnBuilder.cpp:157-169
        add(buildConvolutionalLayer("pool1_features.conv1", 28, 28, 1, 1, 1, 5, 5, 32, th_tiny, in));
        add(buildAvgPoolingLayer("pool1", 24, 24, 32, 2, 2, 2, 2));
        add(buildConvolutionalLayer("pool2_features.conv2", 12, 12, 32, 1, 1, 5, 5, 64, th_tiny * 2, in));
        add(buildAvgPoolingLayer("pool2", 8, 8, 64, 2, 2, 2, 2));
        add(buildFullyConnectedLayer("classifier.fc3", 4 * 4 * 64, 512, th_tiny2, in));
        add(buildFullyConnectedLayer("classifier.fc4", 512, 10, th_tiny2, in));
    } else if (model == "ApproxPlainModel" || model == "PlainModelWoPad") {   // cnnBuilder.cpp:115-134 / :136-155
        const bool avg = model == "ApproxPlainModel";
        add(buildConvolutionalLayer("pool1_features.conv1", 28, 28, 1, 2, 2, 5, 5, 20, th_count, in));
        add(avg ? (Layer *)buildAvgPoolingLayer("pool1", 12, 12, 20, 1, 1, 2, 2) : (Layer *)buildPoolingLayer("pool1", 12, 12, 20, 1, 1, 2, 2));
        add(buildBatchNormLayer("pool1_features.norm1", 20, in));
        add(buildConvolutionalLayer("pool2_features.conv2", 11, 11, 20, 2, 2, 3, 3, 50, avg ? th_count2 : th_count, in));
        add(buildSquareLayer("act1", avg ? th_count2 : th_count));
        add(avg ? (Layer *)buildAvgPoolingLayer("pool2", 5, 5, 50, 1, 1, 2, 2) : (Layer *)buildPoolingLayer("pool2", 5, 5, 50, 1, 1, 2, 2));
        add(buildBatchNormLayer("pool2_features.norm2", 50, in));
        add(buildFullyConnectedLayer("classifier.fc3", 4 * 4 * 50, 500, th_count, in));
        add(buildFullyConnectedLayer("classifier.fc4", 500, 10, avg ? th_count2 : th_count, in));
    } else throw invalid_argument("unknown model " + model);
    return net;
}
Network CnnBuilder::buildAndSaveNetwork(string file_name)
{   // cnnBuilder.cpp:181-196
    ofstream outfile(file_name, ofstream::binary);
    Network net = buildNetwork();
    for (int i = 0; i < net.getNumLayers(); i++) net.getLayer(i)->savePlaintextParameters(&outfile);
    outfile.close();
    return net;
}
