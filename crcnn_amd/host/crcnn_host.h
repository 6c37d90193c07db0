// crcnn_host.h -- C++ host side of the engine: the reference's layer / network / builder interface, same names, same
// constructor argument order, same error behaviour (exceptions), implemented purely on the C ABI of include/crcnn_hip.h.
//
// What differs from CrCNN by design (MI355X-first):
//   * `ciphertext3D` is a handle to a device-resident tensor of ciphertexts [B][z][x][y] (B = image batch, 1 for the
//     reference's single-image calls) instead of nested std::vectors of SEAL objects (CrCNN/src/globals.h:10-16);
//     copying the handle is cheap, layers never modify their argument.
//   * `Plaintext` keeps the sparse balanced-ternary coefficients of an encoded weight (<= 96 non-zeros) instead of a dense
//     n+1 word array; its save/load wire format is SEAL's (plaintext.cpp:346-363) so encoded-model files interchange.
//   * th_count arguments are accepted and ignored (parallelism is the GPU's).
#pragma once
#include <cstdint>
#include <istream>
#include <memory>
#include <ostream>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>
#include "../../include/crcnn_hip.h"

// ---- plaintext / tensor types (CrCNN/src/globals.h:10-16) -----------------------------------------------------------
class Plaintext {
public:
    int coeff_count_ = 0;                                   // SEAL's Plaintext::coeff_count()
    std::vector<std::pair<int, uint64_t>> nz;               // (index, coefficient) for non-zero coefficients
    int coeff_count() const { return coeff_count_; }
    bool is_zero() const { return nz.empty(); }
    void save(std::ostream &stream) const;                  // SEAL wire format: int32 coeff_count, then uint64 coefficients
    void load(std::istream &stream);
    void dense(uint64_t *out, int n) const;                 // zero-extended to n coefficients
};
typedef std::vector<std::vector<std::vector<Plaintext>>> plaintext3D;
typedef std::vector<std::vector<Plaintext>> plaintext2D;
typedef std::vector<std::vector<std::vector<std::vector<Plaintext>>>> plaintext4D;
typedef std::vector<std::vector<std::vector<std::vector<float>>>> floatHypercube;
typedef std::vector<std::vector<std::vector<float>>> floatCube;

struct DeviceBuffer { void *ptr = nullptr; size_t bytes = 0; DeviceBuffer(size_t b); ~DeviceBuffer(); DeviceBuffer(const DeviceBuffer &) = delete;
    DeviceBuffer &operator=(const DeviceBuffer &) = delete; };

class ciphertext3D {          // device tensor of size-2 ciphertexts, [B][zd][xd][yd]
public:
    int B = 0, zd = 0, xd = 0, yd = 0, form = CRC_COEFF;
    std::shared_ptr<DeviceBuffer> buf;
    size_t offset = 0;        // bytes into buf: images() hands out views of a batch without copying it
    ciphertext3D() {}
    ciphertext3D(int B, int zd, int xd, int yd, int form = CRC_COEFF);
    size_t count() const { return (size_t)B * zd * xd * yd; }
    uint64_t *data() const { return buf ? (uint64_t *)((char *)buf->ptr + offset) : nullptr; }
    ciphertext3D images(int b0, int count) const;             // view of images [b0, b0 + count) (ciphertext forms only)
    // CrCNN code indexes input[0].size() etc.; the equivalents:
    int size() const { return zd; }
    static ciphertext3D fromHost(const uint64_t *h, int B, int zd, int xd, int yd);    // h: [B][zd][xd][yd][2][k][n]
    std::vector<uint64_t> toHost() const;
};
ciphertext3D stackImages(const std::vector<ciphertext3D> &images);      // B=1 tensors -> one batch
ciphertext3D deepCopyImage(const ciphertext3D &image);                  // globals.cpp:159-171

// ---- process-global crypto context (CrCNN/src/globals.h:18-48) --------------------------------------------------------
extern crc_ctx *context;                                    // the engine context (SEALContext + Evaluator tables)
// the HIP stream (hipStream_t behind void*; crc_stream_create makes one) all layer calls, copies and synchronisations of these classes go to; NULL = the
// default stream.  Install it before the first forward(); the caller keeps ownership and orders its own streams against it with events
// Announce the number of images a Network::forward will get (0 = unknown, the default) BEFORE the layers' weights are placed (fuse(), broadcastParameters(),
// the first forward()): a deployment that evaluates one image at a time keeps every dense layer's canonical weights resident and streams them, where a batched
// one may drop them for the matrix-core form (PlainModelWoPad's fc3 at n = 16384 has room for one of the two)
void setExpectedBatch(int images_per_forward);
void setStream(void *stream);
void *getStream();
extern std::vector<uint64_t> secret_key, public_key, ev_keys16_host;
extern std::shared_ptr<DeviceBuffer> ev_keys16;             // evaluation keys, dbc = 16, resident in HBM
// Client-side randomness (secret key, evaluation keys, every encryption).  The reference draws from std::random_device (SEAL 2.3.1
// randomgen.cpp:7); here setParameters() draws a fresh 256-bit ChaCha20 key from the OS (crc_random_key -> getrandom(2)) on every
// call and each encryption uses its own keystream under it.  setDeterministicSeed() replaces that by a PUBLIC 64-bit seed so that
// tests, benchmarks and golden vectors are reproducible -- a seeded run is NOT secure (anyone who knows the seed can decrypt).
void setDeterministicSeed(uint64_t seed);                   // tests / bench only; takes effect at the next setParameters()
void clearDeterministicSeed();                              // back to OS entropy (the default)
void setParameters(int poly_modulus = 4096, uint64_t plain_modulus = 1 << 20);          // coeff_modulus_128(poly_modulus)
void setParameters(int poly_modulus, const std::vector<uint64_t> &coeff_modulus, uint64_t plain_modulus, int device = 0);
void delParameters();
// key / ciphertext files in SEAL's wire formats, interchangeable with CrCNN's (globals.cpp:58-111, 174-205)
void setAndSaveParameters(std::string public_key_path, std::string secret_key_path, std::string evaluation_key_path, int poly_modulus, uint64_t plain_modulus);
void initFromKeys(std::string public_key_path, std::string secret_key_path, std::string evaluation_key_path, int poly_modulus, uint64_t plain_modulus);
ciphertext3D encryptAndSaveImage(std::vector<float> image, int zd, int xd, int yd, std::string file_name);
ciphertext3D loadEncryptedImage(int zd, int xd, int yd, std::string file_name);
Plaintext fraencode(double value);                          // fraencoder->encode(value)
double fradecode(const std::vector<uint64_t> &plain);
ciphertext3D encryptImage(std::vector<float> image, int zd, int xd, int yd);             // globals.cpp:127-142
ciphertext3D encryptImage(floatCube image);                                              // globals.cpp:144-157
std::vector<floatCube> decryptImages(const ciphertext3D &encrypted);                     // one floatCube per image of the batch
floatCube decryptImage(const ciphertext3D &encrypted_image);                             // globals.cpp:207-230 (B must be 1)
int noiseBudget(const ciphertext3D &t, size_t index = 0);
// decryptImage -> encryptImage for every image of a batch (the refresh Network::forward runs in front of layer_before_reenc, network.cpp:30-34), on the
// device and on the launch stream: the tensor may be in coefficient or NTT form, comes back in `out_form` (CRC_COEFF / CRC_NTT) under fresh randomness, and
// `values` (optional) receives the floats the client saw, [B][zd][xd][yd] -- asking for them makes the call wait for the stream
ciphertext3D refreshImages(const ciphertext3D &encrypted, int out_form = CRC_COEFF, std::vector<float> *values = nullptr);

// ---- layers (CrCNN/src/layer.h:10-31) ------------------------------------------------------------------------------
class Layer {
public:
    std::string name;
    int out_form = CRC_COEFF;                               // CRC_NTT keeps the output NTT-resident (set by Network::forward)
    Layer() {}
    Layer(std::string layer_name) : name(layer_name) {}
    virtual ~Layer() {}
    std::string getName() { return name; }
    virtual void printLayerStructure() = 0;
    virtual ciphertext3D forward(ciphertext3D input) = 0;
    virtual void savePlaintextParameters(std::ostream *outfile) = 0;
    virtual void loadPlaintextParameters(std::istream *infile) = 0;
    void computeBoundaries(int xd, int yd, int xs, int ys, int xf, int yf, int *xl, int *yl);   // layer.cpp:12-26
    // device-resident encoded parameters of this layer, in a fixed order (Network::broadcastParameters).  allocate_only: a receiving
    // rank sizes the buffers without encoding anything; otherwise the plaintext parameters are lifted + NTT'd into them first.
    virtual void deviceParameters(std::vector<std::shared_ptr<DeviceBuffer>> &out, bool allocate_only) { (void)out; (void)allocate_only; }
    // where this layer keeps its weights: 0 resident in NTT form, 1 streamed (coefficient-form plaintexts, transformed tile by tile inside every forward),
    // 2 tile-wise limb weights (no canonical copy is ever made).  Decided from the memory the rank has; Network::broadcastParameters makes every rank adopt the
    // root's decision, because the buffers on the wire are sized by it
    virtual int placement() { return 0; }
    virtual void adoptPlacement(int p) { (void)p; }
    // reporting (bench_host): bytes this layer holds in HBM right now (parameters in whatever operand form they are in, streaming tiles), and the
    // multiply-accumulate kernel a conv / dense layer runs on ("" for the others)
    virtual size_t deviceBytes() const { return 0; }
    virtual std::string kernelName() const { return ""; }
};

class BatchNormLayer;
class ConvolutionalLayer : public Layer {                   // convolutionalLayer.h:33-34
public:
    friend class Network;
    int xd, yd, zd, xs, ys, xf, yf, nf, th_count;
    int xo, yo, zo;
    plaintext4D filters;                                    // nf,zd,xf,yf
    std::vector<Plaintext> biases;
    bool filters_already_ntt = false;
    ConvolutionalLayer(std::string name, int xd, int yd, int zd, int xs, int ys, int xf, int yf, int nf, int th_count, plaintext4D &filters,
        std::vector<Plaintext> &biases);
    ConvolutionalLayer(std::string name, int xd, int yd, int zd, int xs, int ys, int xf, int yf, int nf, int th_count, std::istream *infile);
    ciphertext3D forward(ciphertext3D input) override;
    plaintext3D getKernel(int kernel_index) { return filters[kernel_index]; }
    Plaintext getBias(int bias_index) { return biases[bias_index]; }
    void savePlaintextParameters(std::ostream *outfile) override;
    void loadPlaintextParameters(std::istream *infile) override;
    void printLayerStructure() override;
private:
    std::shared_ptr<DeviceBuffer> d_w, d_b[2];              // NTT-form weights, bias delta in coefficient / NTT form
    // weights whose NTT form (k rows each) would take more than 75 % of HBM stay coefficient-form plaintexts (ONE row each) and are lifted + transformed a
    // ~2-GiB filter tile at a time inside every forward (SURVEY section 7's fall-back; PlainModelWoPad's fc3 with all eight primes of n = 16384 is 419 GB)
    bool streamed = false;
    int stream_form = CRC_NTT;                              // operand form of the last streamed forward (CRC_NTTL: 64-filter limb tiles on the matrix cores)
    std::shared_ptr<DeviceBuffer> d_plain, d_wtile, d_ytile;
    // CRC_NTTP / CRC_NTTL / CRC_NTTL1 once Network::forward has put the weights into their MAC kernel's operand form
    int w_form = CRC_NTT;
    std::shared_ptr<DeviceBuffer> d_w_canon;                // CRC_NTTL1 only: the canonical NTT-form weights
    int forced_placement = -1;
    void upload();
    void packWeights(bool unpack);
    // -> CRC_NTTL (matrix-core kernel) when the layer qualifies (for batches of B) and HBM has room for the second copy
    bool limbWeights(int B);
public:
    void deviceParameters(std::vector<std::shared_ptr<DeviceBuffer>> &out, bool allocate_only) override;
    int placement() override;
    void adoptPlacement(int p) override { forced_placement = p; }
    // weights back to canonical NTT form (unpacked; rebuilt from the plaintexts when the matrix-core form replaced them)
    void restoreCanonical();
    size_t deviceBytes() const override;
    std::string kernelName() const override;
};

class FullyConnectedLayer : public Layer {                  // fullyConnectedLayer.h:22-24
public:
    friend class Network;
    int in_dim, out_dim, th_count;
    plaintext2D weights;
    std::vector<Plaintext> biases;
    bool weights_already_ntt = false;
    FullyConnectedLayer(std::string name, int in_dim, int out_dim, int th_count, plaintext2D &weights, std::vector<Plaintext> &biases);
    FullyConnectedLayer(std::string name, int in_dim, int out_dim, int th_count, std::istream *infile);
    ciphertext3D forward(ciphertext3D input) override;
    Plaintext getWeight(int x_index, int y_index) { return weights[x_index][y_index]; }
    Plaintext getBias(int x_index) { return biases[x_index]; }
    void savePlaintextParameters(std::ostream *outfile) override;
    void loadPlaintextParameters(std::istream *infile) override;
    void printLayerStructure() override;
private:
    std::shared_ptr<DeviceBuffer> d_w, d_b[2];
    bool streamed = false;                                  // see ConvolutionalLayer
    int stream_form = CRC_NTT;
    std::shared_ptr<DeviceBuffer> d_plain, d_wtile, d_ytile;
    int w_form = CRC_NTT;
    // A layer whose canonical NTT-form weights and their limb copy do not fit in HBM together (PlainModelWoPad's fc3 at n = 16384, k = 4: 202 + 182 GiB) never
    // gets a canonical copy: its limb weights are built a tile of output rows at a time straight from the plaintexts (lift + NTT -> batch-norm fold of the tile
    // -> pack), a batch-norm layer that Network::fuse() folds into it being applied to every tile (same ciphertexts; netrun.py does the same)
    bool tilewise = false, tile_built = false;
    int last_B = 0;                                         // images of the last forward (kernelName: one image runs as a weight stream)
    int forced_placement = -1;
    std::shared_ptr<BatchNormLayer> fold_bn;
    void buildTilewise();
    void upload();
    void packWeights(bool unpack);
    bool limbWeights(int B);
    bool streamsOnMatrixCores(int B);                      // streamed, and a launch on B images takes the limb GEMM (forwardStreamed's 64-filter limb tiles)
public:
    void deviceParameters(std::vector<std::shared_ptr<DeviceBuffer>> &out, bool allocate_only) override;
    int placement() override;
    void adoptPlacement(int p) override { forced_placement = p; }
    void restoreCanonical();
    size_t deviceBytes() const override;
    std::string kernelName() const override;
};

class PoolingLayer : public Layer {                         // poolingLayer.h:15
public:
    friend class Network;
    int xd, yd, zd, xs, ys, xf, yf, xo, yo, zo;
    PoolingLayer(std::string name, int xd, int yd, int zd, int xs, int ys, int xf, int yf);
    ciphertext3D forward(ciphertext3D input) override;
    void savePlaintextParameters(std::ostream *) override {}
    void loadPlaintextParameters(std::istream *) override {}
    void printLayerStructure() override;
protected:
    std::shared_ptr<DeviceBuffer> d_div;                    // NTT-form divisor (AvgPoolingLayer only)
public:
    size_t deviceBytes() const override { return d_div ? d_div->bytes : 0; }
};

class AvgPoolingLayer : public PoolingLayer {               // avgPoolingLayer.h:11
public:
    Plaintext div_factor;
    AvgPoolingLayer(std::string name, int xd, int yd, int zd, int xs, int ys, int xf, int yf);
};

class SquareLayer : public Layer {                          // squareLayer.h:12
public:
    int th_count;
    SquareLayer(std::string name, int th_count) : Layer(name), th_count(th_count) {}
    ciphertext3D forward(ciphertext3D input) override;
    void savePlaintextParameters(std::ostream *) override {}
    void loadPlaintextParameters(std::istream *) override {}
    void printLayerStructure() override;
private:
};

// Network::fuse(): a SquareLayer with a (sum or average) PoolingLayer behind it.  Relinearisation is linear in the digit polynomials of c2, so the digits of a
// pooling window are added and ONE key switch serves the pooled ciphertext (crc_square_pool_relin_forms): the ciphertexts squareLayer.cpp:21-39 followed by
// poolingLayer.cpp:22-44 produce, bit for bit, with xo yo / (xd yd) of the key-switching work
class SquarePoolLayer : public Layer {
public:
    int xd, yd, zd, xs, ys, xf, yf, xo, yo, zo, th_count;
    SquarePoolLayer(std::string name, int xd, int yd, int zd, int xs, int ys, int xf, int yf, int th_count, std::shared_ptr<DeviceBuffer> d_div);
    ciphertext3D forward(ciphertext3D input) override;
    void savePlaintextParameters(std::ostream *) override {}
    void loadPlaintextParameters(std::istream *) override {}
    void printLayerStructure() override;
    size_t deviceBytes() const override { return d_div ? d_div->bytes : 0; }
private:
    std::shared_ptr<DeviceBuffer> d_div;                    // NTT-form divisor of an average pooling (applied to the pooled ciphertexts)
};

class BatchNormLayer : public Layer {                       // batchNormLayer.h:18-20
public:
    friend class Network;
    friend class FullyConnectedLayer;
    int num_channels;
    std::vector<Plaintext> mean, var;                       // var already holds encode(1/sqrt(var+1e-5)) (cnnBuilder.cpp:100-102)
    BatchNormLayer(std::string name, int num_channels, std::vector<Plaintext> &mean, std::vector<Plaintext> &var);
    BatchNormLayer(std::string name, int num_channels, std::istream *infile);
    ciphertext3D forward(ciphertext3D input) override;
    Plaintext getMean(int index) { return mean[index]; }
    Plaintext getVar(int index) { return var[index]; }
    void savePlaintextParameters(std::ostream *outfile) override;
    void loadPlaintextParameters(std::istream *infile) override;
    void printLayerStructure() override;
private:
    std::shared_ptr<DeviceBuffer> d_mean[2], d_invstd;
public:
    size_t deviceBytes() const override { return (d_mean[0] ? d_mean[0]->bytes : 0) + (d_mean[1] ? d_mean[1]->bytes : 0) + (d_invstd ? d_invstd->bytes : 0); }
private:
    void upload();
public:
    void deviceParameters(std::vector<std::shared_ptr<DeviceBuffer>> &out, bool allocate_only) override;
};

// ---- network (CrCNN/src/network.h:11-39) ---------------------------------------------------------------------------
class OutOfBudgetException : public std::exception {
public:
    const int last_layer_computed;
    std::string msg;
    OutOfBudgetException(int last_layer_computed) : last_layer_computed(last_layer_computed), msg("OutOfBudgetException at layer " +
        std::to_string(last_layer_computed)) {}
    const char *what() const throw() override { return msg.c_str(); }
};

class Network {
public:
    std::vector<std::shared_ptr<Layer>> layers;
    // network.cpp:23 hard-codes a client-side decrypt/re-encrypt "refresh" before layer 6; it needs the secret key and is off
    // the accelerated path, so it is a setting here: 6 reproduces the committed reference, -1 (default) never refreshes.
    int layer_before_reenc = -1;
    bool ntt_resident = true;                               // keep tensors in NTT form between linear layers (bit-identical)
    // conv / dense layers with long reductions (>= 8 steps of 32 channels) run as an int8 limb GEMM on the matrix cores (CRC_NTTL, kernels_mfma.hip):
    // exact integer arithmetic, identical ciphertexts, about 4x the vector-ALU kernel.  The conversion of a layer's weights drops their canonical copy:
    // broadcastParameters() must come before the first forward(); fuse() may follow one (it rebuilds the canonical weights from the plaintexts).
    bool matrix_cores = true;
    // >= 0 selects the budget-checking forward the reference keeps for its parameter search (network.cpp:52-96): after every layer the
    // noise budget of output[0][0][0] is measured (secret key, coefficient form at every boundary); at <= 5 bits the layer's input is
    // refreshed and the layer repeated while refreshes are left, then OutOfBudgetException(i - 1) is thrown.  -1: plain forward.
    int max_num_of_reencryptions = -1;
    std::vector<double> last_layer_ms;                      // per-layer wall milliseconds of the last forward (T_LAYER_i, mainparams.cpp:81)
    // true: last_layer_ms comes from HIP events recorded on the launch stream around every layer call -- no synchronisation between the layers, what a
    // throughput measurement wants (crcnn_amd/host/bench_host.cpp); false: wall clock around Layer::forward + a stream synchronisation, as the reference's
    // driver measures
    bool time_with_events = false;
    // Layer::forward calls per layer in the last forward (two-level chunking: a head layer runs once per chunk)
    std::vector<int> last_layer_launches;
    double last_reenc_ms = 0.0;
    // true: the floats the client saw at the refresh(es) of the last forward are kept (in the order the refreshes ran: chunk by chunk under two-level
    // chunking); costs a stream synchronisation per refresh
    bool keep_reenc_values = false;
    std::vector<float> last_reenc_values;
    // Two-level chunking (> 0): the layers in front of the first dense layer run on sub-batches of `head_chunk` images, the dense layers once on the whole
    // batch -- a dense layer streams all of its weights per launch, so its time per image falls with the rows it is used for (PlainModelWoPad at n = 16384:
    // 6-image chunks fit beside 190 GiB of weights, fc3 wants 24+ images).  0: every layer on the whole batch
    int head_chunk = 0;
    // where this rank's HBM goes (bytes): the layers' parameters in their current operand forms, the activation slots forward() keeps, the shared work buffer,
    // the evaluation keys
    struct HbmPlan { size_t parameters = 0, activations = 0, work = 0, keys = 0; };
    HbmPlan hbmPlan() const;
private:
    struct EventPool { std::vector<void *> ev; ~EventPool(); };
    std::shared_ptr<EventPool> event_pool;                  // HIP events of time_with_events, reused from forward to forward (copies of a Network share them)
public:
    std::shared_ptr<DeviceBuffer> tail_slot;                // ... and the dense layers' whole-batch input under two-level chunking
    // the two ping-pong activation buffers forward() keeps across calls (sized by the largest layer output so far)
    std::shared_ptr<DeviceBuffer> act_slot[2];
    Network() {}
    ~Network() {}
    int getNumLayers() { return (int)layers.size(); }
    virtual std::shared_ptr<Layer> getLayer(int i) { return layers[i]; }
    std::vector<std::shared_ptr<Layer>> &getLayers() { return layers; }
    void printNetworkStructure();
    ciphertext3D forward(ciphertext3D input);
    // Exact layer folding (ring algebra over Z_q), done once on the device-resident parameters: a sum/avg pooling
    // layer is folded into the convolution in front of it (pooled kernel, xf' = (pxf-1)*cxs + xf, stride cxs*pxs) when that is
    // estimated to be cheaper, a batch-norm layer into the conv / dense layer behind it (w' = w (*) s[channel],
    // b' = b - sum_taps w' (*) mean[channel]).  The network's output ciphertexts stay bit-identical; only the folded layers'
    // intermediate tensors disappear (their plaintext parameters can no longer be saved).  Returns the number of layers removed.
    int fuse();
    // Multi-GPU start-up (no analogue in the reference): one process (or host thread) per GPU, images sharded across them
    // with no data-path collective.  Rank `root` holds the encoded model -- this call lifts + NTTs its plaintext parameters if that has
    // not happened yet -- and every other rank receives the NTT-form weights / bias / batch-norm rows and the evaluation keys over RCCL
    // (crc_broadcast_weights: ncclBroadcast in <= 1 GiB pieces), without encoding anything.  Every rank then checksums what it holds
    // (crc_checksum64) and the sums are compared with the root's; a mismatch throws std::runtime_error.  Call it before fuse() and
    // before the first forward() on the receiving ranks.  Returns the bytes received per rank.
    // encode_locally = true is SURVEY 8e's alternative: nothing but the evaluation keys is sent, every rank encodes + transforms the weights itself from the
    // model file it read (the 2 MB of floats instead of 35-200 GB of residues on the wire; the same placement agreement and the same checksum comparison).
    size_t broadcastParameters(crc_comm *comm, int root = 0, bool encode_locally = false);
};

// ---- model loader + builder (CrCNN/src/cnnBuilder.h:16-44) ----------------------------------------------------------
class CnnBuilder {
public:
    std::string plain_model_path;
    CnnBuilder(std::string plain_model_path) : plain_model_path(plain_model_path) {}
    ~CnnBuilder() {}
    std::vector<float> getPretrained(std::string var_name);
    ConvolutionalLayer *buildConvolutionalLayer(std::string name, int xd, int yd, int zd, int xs, int ys, int xf, int yf, int nf, int th_count,
        std::istream *infile);
    FullyConnectedLayer *buildFullyConnectedLayer(std::string name, int in_dim, int out_dim, int th_count, std::istream *infile);
    PoolingLayer *buildPoolingLayer(std::string name, int xd, int yd, int zd, int xs, int ys, int xf, int yf);
    AvgPoolingLayer *buildAvgPoolingLayer(std::string name, int xd, int yd, int zd, int xs, int ys, int xf, int yf);
    SquareLayer *buildSquareLayer(std::string name, int th_count);
    BatchNormLayer *buildBatchNormLayer(std::string name, int num_channels, std::istream *infile);
    // cnnBuilder.cpp:108-179 hard-codes one topology per source edit (Tiny is the committed one); all three are available here
    Network buildNetwork(std::string file_name = "");                  // PlainModelTiny, as committed (cnnBuilder.cpp:157-169)
    Network buildNetworkByName(const std::string &model, std::string file_name = "");   // "PlainModelTiny" | "ApproxPlainModel" | "PlainModelWoPad"
    Network buildAndSaveNetwork(std::string file_name);
};
