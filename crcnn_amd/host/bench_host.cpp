// bench_host.cpp -- the MEASURED path of bench.py: the C++ host classes (crcnn_host.h: the drop-in for the reference's Layer / Network / CnnBuilder) over the C
// ABI. The reference's timed driver is CrCNN/src/mainparams.cpp:64-116 (one image at a time, chrono around every layer -> the T_LAYER_i columns,
// mainparams.cpp:81); this is the same loop over chunks of encrypted images: CnnBuilder builds the network from the HDF5 model, Network::fuse() folds it,
// Network::forward runs it.
//
//   bench_host model=<name> h5=<model.h5> n=<n> k=<k> t=<t> [q=<p0>,<p1>,..] inputs=<file> distinct=<D> batch=<B> chunk=<C> group=<G> steps=<K> warmup=<W>
//              outputs=<file> [fuse=1] [matrix_cores=1] [reenc=<layer>]
//
// inputs        D encrypted images ([D][784][2][k][n] u64, coefficient form: bench.py's client side writes them), tiled to one launch of C * G images that
//               every chunk of the batch re-reads (the same bytes per image as a resident batch; bench.py states it under "data")
// group         > 1: two-level chunking -- Network::forward gets C * G images with head_chunk = C: the layers in front of the first dense layer per chunk, the
//               dense layers once per group
// timing        W untimed passes over the batch, then K timed ones bracketed by stream synchronisations; per-layer times from HIP events on the launch stream
//               (Network::time_with_events: no synchronisation between layers)
// outputs       the 10 output ciphertexts of the first D images of an untimed launch ([D][10][2][k][n] u64): bench.py hashes them against the reference's
//               goldens and decrypts them
// matrix_cores  0: Network::matrix_cores = false -- every conv / dense layer on the vector-ALU kernel (mac3_kernel) and the row NTT: north_star's "no MFMA" path
// reenc         >= 0: Network::layer_before_reenc (counted on the unfused network, as network.cpp:23 counts): the client-side refresh of network.cpp:30-34, on
//               the device (refreshImages), under this process' seeded client keys; its time is reported as T_REENC (mainparams.cpp:81).  The re-encryption is
//               randomised, so launches are compared through their DECRYPTED outputs
// Several ranks (world=<N> rank=<r> rendezvous=<file>; one process per GPU, started by bench.py -- the reference's driver has no counterpart:
//               mainparams.cpp:64-116 runs one process): rank 0 makes the RCCL rendezvous id (crc_comm_unique_id) and writes it to <file>, every rank joins
//               (crc_comm_create), the encoded model goes out once with Network::broadcastParameters (the ONE collective of the path), then every rank
//               evaluates its own batch: no data-path collective.  The timed region is bracketed on every rank by an all-gather (the barrier) + stream
//               synchronisation, the elapsed times are gathered and the line carries the MAX over ranks and the whole-job rate (all ranks' images / that time).
// weights_via   (world > 1) broadcast: rank 0 encodes + transforms the weights and Network::broadcastParameters sends them (default); floats: every rank
//               encodes + transforms its own copy from the model file's floats, nothing but the evaluation keys' checksum crosses the wire (SURVEY 8e's
//               "cheaper alternative to measure against")
// sync_each     1: the stream is synchronised after every Network::forward and the wall time of forward + wait is reported (ms_per_image_sync_each): with
//               batch=1 chunk=1 the single-image latency of mainparams.cpp:85-112's usage (one image at a time)
// launch_check  1: rendezvous through the file only (no GPU, no RCCL): the CPU-side test of the launcher
// stream_inputs ciphertext|plaintext: the input launch is not resident -- see stream_inputs below
// Prints ONE JSON line on stdout.
#include "crcnn_host.h"
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iostream>
#include <map>
#include <thread>
#include <cerrno>
#include <fcntl.h>
#include <unistd.h>
using namespace std;

// rank 0 writes the 128-byte id to `path` (temporary name + rename: readers never see half a file), the others wait for it
static void rendezvous_id(const string &path, int rank, uint8_t *id, bool make_nccl_id)
{
    if (rank == 0) {
        if (make_nccl_id) { if (crc_comm_unique_id(id)) throw runtime_error("crc_comm_unique_id failed (rccl error " + to_string(crc_last_comm_error()) +
            ")"); }
        else { ifstream r("/dev/urandom", ios::binary); r.read((char *)id, CRC_COMM_ID_BYTES); }
        // a fresh file nobody else can read, never through a symlink somebody planted (O_EXCL | O_NOFOLLOW, mode 0600); bench.py puts it into a 0700 directory
        const string tmp = path + ".tmp";
        const int fd = open(tmp.c_str(), O_WRONLY | O_CREAT | O_EXCL | O_NOFOLLOW | O_CLOEXEC, 0600);
        if (fd < 0) throw runtime_error("cannot create " + tmp + ": " + strerror(errno));
        const bool wrote = write(fd, id, CRC_COMM_ID_BYTES) == CRC_COMM_ID_BYTES;
        if (close(fd) || !wrote) throw runtime_error("cannot write " + tmp);
        if (rename(tmp.c_str(), path.c_str())) throw runtime_error("cannot rename " + tmp);
        return;
    }
    for (int tries = 0; tries < 6000; tries++) {               // 10 minutes: rank 0 encrypts its inputs before it starts its bench_host
        ifstream f(path, ios::binary);
        if (f && f.read((char *)id, CRC_COMM_ID_BYTES)) return;
        this_thread::sleep_for(chrono::milliseconds(100));
    }
    throw runtime_error("rank " + to_string(rank) + ": no rendezvous id in " + path + " after 10 minutes");
}

static string jstr(const string &s) { string o = "\""; for (char c : s) { if (c == '"' || c == '\\') o += '\\'; o += c; } return o + "\""; }

int main(int argc, char **argv)
{
    map<string, string> a;
    for (int i = 1; i < argc; i++) {
        const char *eq = strchr(argv[i], '=');
        if (!eq) { fprintf(stderr, "bench_host: argument without '=': %s\n", argv[i]); return 1; }
        a[string(argv[i], (size_t)(eq - argv[i]))] = eq + 1;
    }
    auto need = [&](const char *k) -> string { if (!a.count(k)) { fprintf(stderr, "bench_host: missing %s=\n", k); exit(1); } return a[k]; };
    auto geti = [&](const char *k, long long d) { return a.count(k) ? atoll(a[k].c_str()) : d; };
    const int rank = (int)geti("rank", 0), world = max(1, (int)geti("world", 1));
    // the ranks of a node share its host cores: the item loops of the engine (weight encoding, tile-wise limb weights) take cores / ranks threads each
    if (world > 1 && !getenv("CRC_LOCAL_WORLD")) setenv("CRC_LOCAL_WORLD", to_string((int)geti("local_world", world)).c_str(), 1);
    if (geti("launch_check", 0)) {                              // CPU-side check of the launcher: the file rendezvous and the thread share, nothing else
        try {
            uint8_t id[CRC_COMM_ID_BYTES];
            rendezvous_id(need("rendezvous"), rank, id, false);
            uint64_t h = 1469598103934665603ULL; for (uint8_t b : id) h = (h ^ b) * 1099511628211ULL;
            printf("{\"launch_check\": true, \"host\": \"C++ bench_host\", \"rank\": %d, \"world\": %d, \"id_fnv1a\": \"%016llx\", \"host_threads\": %d}\n",
                rank, world,
                   (unsigned long long)h, crc_host_thread_limit());
            return 0;
        } catch (const exception &e) { fprintf(stderr, "bench_host: %s\n", e.what()); return 2; }
    }
    const string model = need("model"), h5 = need("h5"), inputs = need("inputs"), outputs = need("outputs");
    const int n = (int)geti("n", 0), k = (int)geti("k", 0); const uint64_t t = strtoull(need("t").c_str(), 0, 0);
    const int distinct = (int)geti("distinct", 1), batch = (int)geti("batch", 1), head = (int)geti("chunk", 1), group = max(1, (int)geti("group", 1));
    const int steps = (int)geti("steps", 1), warmup = (int)geti("warmup", 0), fuse = (int)geti("fuse", 1);
    const int reenc = (int)geti("reenc", -1), matrix_cores = (int)geti("matrix_cores", 1);
    const int sync_each = (int)geti("sync_each", 0);
    const string weights_via = a.count("weights_via") ? a["weights_via"] : "broadcast";
    if (weights_via != "broadcast" && weights_via != "floats") { fprintf(stderr, "bench_host: weights_via= broadcast | floats\n"); return 1; }
    const int launch = head * group;                          // images per Network::forward
    // stream_inputs: after the resident measurement, `stream_steps` more passes in which every launch's images come over PCIe while the previous launch is
    // evaluated (mainparams.cpp:85-112 encrypts, evaluates and decrypts image after image).  ciphertext: 784 ciphertexts per image from page-locked host
    // memory; plaintext: the 784 pixel plaintexts per image (n words each) and Encryptor::encrypt on the device (crc_encrypt_dev_forms) inside the pipeline
    const string stream_mode = a.count("stream_inputs") ? a["stream_inputs"] : "none";
    const int stream_steps = (int)geti("stream_steps", 1);
    try {
        if (n < 1 || k < 1 || launch < 1 || batch < launch || distinct < 1 || distinct > launch || steps < 1)
            throw invalid_argument("bad sizes (need 1 <= distinct <= chunk * group <= batch)");
        {
            bool known = false;
            for (const char *m : {"none", "ciphertext", "plaintext", "ciphertext,plaintext", "plaintext,ciphertext"}) known = known || stream_mode == m;
            if (!known) throw invalid_argument("stream_inputs= none | ciphertext | plaintext | ciphertext,plaintext");
        }
        uint64_t q[16];
        // explicit coefficient modulus (q=<prime>,<prime>,...); default: the first k primes of coeff_modulus_128(n)
        if (a.count("q")) {
            int cnt = 0; const char *p = a["q"].c_str();
            while (*p && cnt < 16) { char *e; q[cnt++] = strtoull(p, &e, 0); p = *e == ',' ? e + 1 : e; if (e == p && *e) break; }
            if (cnt != k) throw invalid_argument("q= must list k primes");
        } else {
            const int kd = crc_default_coeff_modulus_128(n, q, 16);
            if (kd < k) throw invalid_argument("coeff_modulus_128(n) has fewer primes than asked for");
        }
        const auto t_setup = chrono::high_resolution_clock::now();
        // the seeded client side of bench.py: the same evaluation keys, hence the same ciphertexts behind Square
        setDeterministicSeed((uint64_t)geti("key_seed", 2024));
        setParameters(n, vector<uint64_t>(q, q + k), t, (int)geti("device", 0));
        // every call of the host classes goes to ONE non-blocking stream of this process (no implicit ordering against the copy stream of the streamed mode)
        void *compute = nullptr, *copy = nullptr;
        if (crc_stream_create(context, &compute) || crc_stream_create(context, &copy)) throw runtime_error("crc_stream_create");
        setStream(compute);
        size_t free0 = 0, total = 0;
        crc_mem_info(context, &free0, &total);
        // ---- several ranks: communicator, then the one collective of the path
        crc_comm *comm = nullptr;
        vector<uint64_t> gathered((size_t)world * 4);
        // (also the barrier: it synchronises the stream and waits for every rank)
        auto allgather = [&](uint64_t v0, uint64_t v1, uint64_t v2, uint64_t v3) {
            uint64_t mine[4] = {v0, v1, v2, v3};
            if (!comm) { memcpy(gathered.data(), mine, sizeof mine); return; }
            if (crc_comm_allgather_u64(comm, mine, 4, gathered.data(), compute)) throw runtime_error("crc_comm_allgather_u64 failed (rccl error " +
                to_string(crc_last_comm_error()) + ")");
        };
        if (world > 1) {
            uint8_t id[CRC_COMM_ID_BYTES];
            rendezvous_id(need("rendezvous"), rank, id, true);
            // (tests: CRC_TEST_BREAK_COMM_RANK=<r> spoils rank r's copy of the id, so that its crc_comm_create fails while the others wait for it)
            if (getenv("CRC_TEST_BREAK_COMM_RANK") && atoi(getenv("CRC_TEST_BREAK_COMM_RANK")) == rank) for (int i = 8; i < 40; i++) id[i] ^= 0x15;
            if (crc_comm_create(context, world, rank, id, &comm)) throw runtime_error("crc_comm_create failed (rccl error " +
                to_string(crc_last_comm_error()) + ")");
        }
        setExpectedBatch(launch);                             // (one image per forward: dense weights stay canonical and are streamed)
        CnnBuilder build(h5);
        Network net = build.buildNetworkByName(model);
        double bcast_s = 0.0; size_t bcast_bytes = 0;
        if (comm) {
            allgather(0, 0, 0, 0);
            const auto tb = chrono::high_resolution_clock::now();
            // rank 0 lifts + transforms the weights, everybody else receives them (checksums compared inside)
            bcast_bytes = net.broadcastParameters(comm, 0, weights_via == "floats");
            allgather(0, 0, 0, 0);
            bcast_s = chrono::duration<double>(chrono::high_resolution_clock::now() - tb).count();
        }
        net.layer_before_reenc = reenc;                       // (before fuse(): the index follows the layers it counts)
        net.matrix_cores = matrix_cores != 0;
        if (fuse) net.fuse();
        if (group > 1) net.head_chunk = head;
        net.time_with_events = true;
        // one launch's images: the D distinct ones tiled
        const size_t ctw = (size_t)2 * k * n, imgw = 784 * ctw;
        ciphertext3D x;
        vector<uint64_t> h((size_t)distinct * imgw);
        {
            ifstream f(inputs, ios::binary); if (!f) throw runtime_error("cannot open " + inputs);
            f.read((char *)h.data(), (streamsize)(h.size() * 8)); if (!f) throw runtime_error("short read: " + inputs);
            vector<ciphertext3D> one, tiled;
            for (int d = 0; d < distinct; d++) one.push_back(ciphertext3D::fromHost(h.data() + (size_t)d * imgw, 1, 1, 28, 28));
            for (int b = 0; b < launch; b++) tiled.push_back(one[b % distinct]);
            x = stackImages(tiled);
        }
        if (stream_mode.find("ciphertext") == string::npos) { h.clear(); h.shrink_to_fit(); }
        // with a refresh in the network every launch draws fresh randomness: two launches are "the same" when their outputs DECRYPT to the same plaintexts
        auto decrypted = [&](const vector<uint64_t> &cts) {
            vector<uint64_t> pl((size_t)distinct * 10 * n);
            if (crc_decrypt(context, secret_key.data(), cts.data(), (size_t)distinct * 10, 2, pl.data())) throw runtime_error("crc_decrypt");
            return pl;
        };
        vector<uint64_t> first_dec;
        // untimed first launch: operand forms, module load; the distinct images' outputs
        {
            ciphertext3D y = net.forward(x);
            vector<uint64_t> yh = y.toHost();
            if (reenc >= 0) first_dec = decrypted(yh);
            ofstream o(outputs, ios::binary); o.write((const char *)yh.data(), (streamsize)((size_t)distinct * 10 * ctw * 8));
            if (!o) throw runtime_error("cannot write " + outputs);
        }
        const double setup_s = chrono::duration<double>(chrono::high_resolution_clock::now() - t_setup).count();
        size_t free1 = 0; crc_mem_info(context, &free1, &total);
        // a rank that starts its timed region with a few GB left dies in the first allocation that grows (an activation slot, the work buffer): say so now,
        // readably
        const double min_free_gib = a.count("min_free_gib") ? atof(a["min_free_gib"].c_str()) : (world > 1 ? 8.0 : 0.0);
        if ((double)free1 < min_free_gib * (double)((size_t)1 << 30))
            throw runtime_error("rank " + to_string(rank) + ": only " + to_string(free1 >> 20) + " MiB of HBM free after the first launch (min_free_gib=" +
                to_string(min_free_gib) +
                                "): lower chunk= / group= or the batch per GPU");
        const Network::HbmPlan plan = net.hbmPlan();
        const int L = net.getNumLayers();
        const int launches = batch / launch;
        for (int s = 0; s < warmup; s++) for (int c = 0; c < launches; c++) net.forward(x);
        vector<double> tl(L, 0.0); vector<long long> calls(L, 0);
        double t_reenc = 0.0;
        double t_sync = 0.0;
        allgather(0, 0, 0, 0);                                    // barrier + stream synchronisation on every rank
        const auto t0 = chrono::high_resolution_clock::now();
        ciphertext3D y_last;
        for (int s = 0; s < steps; s++)
            for (int c = 0; c < launches; c++) {
                const auto tf = chrono::high_resolution_clock::now();
                y_last = net.forward(x);
                if (sync_each) {                                  // latency of ONE launch as a caller sees it: forward + the wait for its last kernel
                    if (crc_stream_sync(context, compute) < 0) throw runtime_error("crc_stream_sync");
                    t_sync += chrono::duration<double, milli>(chrono::high_resolution_clock::now() - tf).count();
                }
                for (int i = 0; i < L; i++) { tl[i] += net.last_layer_ms[i]; calls[i] += net.last_layer_launches[i]; }
                t_reenc += net.last_reenc_ms;
            }
        if (crc_stream_sync(context, compute) < 0) throw runtime_error("crc_stream_sync");
        const double dt = chrono::duration<double>(chrono::high_resolution_clock::now() - t0).count();
        // the LAST timed launch is verified too, outside the timed region: its ciphertexts must be the untimed first launch's (which bench.py checks against the
        // reference) bit for bit
        bool timed_same = false;
        {
            vector<uint64_t> yh = y_last.toHost(), ref((size_t)distinct * 10 * ctw);
            ifstream f(outputs, ios::binary); f.read((char *)ref.data(), (streamsize)(ref.size() * 8));
            timed_same = f && memcmp(yh.data(), ref.data(), ref.size() * 8) == 0;
            if (reenc >= 0) timed_same = f && decrypted(yh) == first_dec && memcmp(yh.data(), ref.data(), ref.size() * 8) != 0;
            y_last = ciphertext3D();
        }
        const double images = (double)steps * launches * launch;
        allgather((uint64_t)(dt * 1e9), (uint64_t)images, (uint64_t)free1, (uint64_t)plan.parameters);
        double dt_max = 0.0, images_all = 0.0;
        for (int r = 0; r < world; r++) { dt_max = max(dt_max, (double)gathered[(size_t)r * 4] * 1e-9); images_all += (double)gathered[(size_t)r * 4 + 1]; }
        const vector<uint64_t> per_rank = gathered;

        // ---- streamed inputs (resident measurement above untouched): double-buffered uploads on the copy stream beside the kernels of the compute stream
        bool st_same = true;
        string streamed_json;
        vector<string> modes;
        if (stream_mode != "none") { size_t p0 = 0; while (p0 <= stream_mode.size()) { const size_t c = stream_mode.find(',', p0);
            modes.push_back(stream_mode.substr(p0, c == string::npos ? c : c - p0)); if (c == string::npos) break; p0 = c + 1; } }
        for (const string &mode : modes) {
            double st_dt = 0.0, st_images = 0.0, st_bytes = 0.0; bool same = true;
            const bool pt = mode == "plaintext";
            const size_t unit = pt ? (size_t)784 * n : imgw;                       // words uploaded per image
            uint64_t *pinned = nullptr;
            if (crc_host_alloc(context, (size_t)distinct * unit * 8, (void **)&pinned))
                throw runtime_error("crc_host_alloc: " + to_string((size_t)distinct * unit * 8) + " bytes of page-locked host memory for the streamed inputs");
            if (pt) {
                ifstream f(need("plain_inputs"), ios::binary); if (!f) throw runtime_error("cannot open plain_inputs");
                f.read((char *)pinned, (streamsize)((size_t)distinct * unit * 8)); if (!f) throw runtime_error("short read: plain_inputs");
            } else {
                memcpy(pinned, h.data(), (size_t)distinct * unit * 8);
                h.clear(); h.shrink_to_fit();                      // one copy of the distinct images in host memory, not two (ciphertext mode runs once)
            }
            // (plaintext mode: the device encryptor leaves NTT-form ciphertexts -- crc_encrypt_dev_forms: three forward transforms per modulus, none back --
            // and the first layer skips the transform it runs on a coefficient-form image)
            const int xform = pt ? CRC_NTT : CRC_COEFF;
            ciphertext3D xin[2] = {ciphertext3D(launch, 1, 28, 28, xform), ciphertext3D(launch, 1, 28, 28, xform)};
            shared_ptr<DeviceBuffer> up[2], d_pk, d_encwork;
            if (pt) {
                for (auto &u : up) u = make_shared<DeviceBuffer>((size_t)launch * unit * 8);
                d_pk = make_shared<DeviceBuffer>(public_key.size() * 8);
                d_encwork = make_shared<DeviceBuffer>(crc_encrypt_dev_work_bytes(context, (size_t)launch * 784));
                if (crc_memcpy_h2d(context, d_pk->ptr, public_key.data(), public_key.size() * 8, compute) || crc_stream_sync(context,
                    compute)) throw runtime_error("public key upload");
            }
            void *copied[2], *consumed[2];
            for (int i = 0; i < 2; i++) if (crc_event_create(context, &copied[i]) || crc_event_create(context,
                &consumed[i])) throw runtime_error("crc_event_create");
            uint64_t enc_seed = 0x5eed0000;
            auto upload = [&](int slot, bool wait_consumed) {
                if (wait_consumed && crc_stream_wait_event(context, copy, consumed[slot])) throw runtime_error("crc_stream_wait_event");
                char *dst = pt ? (char *)up[slot]->ptr : (char *)xin[slot].data();
                for (int b = 0; b < launch; b++)
                    if (crc_memcpy_h2d(context, dst + (size_t)b * unit * 8, pinned + (size_t)(b % distinct) * unit, unit * 8,
                        copy)) throw runtime_error("crc_memcpy_h2d");
                if (crc_event_record(context, copied[slot], copy)) throw runtime_error("crc_event_record");
            };
            auto run = [&](int slot) {
                if (crc_stream_wait_event(context, compute, copied[slot])) throw runtime_error("crc_stream_wait_event");
                if (pt && crc_encrypt_dev_forms(context, (const uint64_t *)d_pk->ptr, (const uint64_t *)up[slot]->ptr, (size_t)launch * 784, enc_seed++,
                    CRC_NTT,
                                                xin[slot].data(), d_encwork->ptr, compute)) throw runtime_error("crc_encrypt_dev_forms");
                ciphertext3D y = net.forward(xin[slot]);
                if (crc_event_record(context, consumed[slot], compute)) throw runtime_error("crc_event_record");
                return y;
            };
            // untimed: one streamed launch, compared with the resident launch's outputs (ciphertext mode: the same ciphertexts in, the same bits out)
            upload(0, false);
            {
                ciphertext3D y = run(0);
                if (!pt) {
                    vector<uint64_t> yh = y.toHost(), ref((size_t)distinct * 10 * ctw);
                    ifstream f(outputs, ios::binary); f.read((char *)ref.data(), (streamsize)(ref.size() * 8));
                    same = f && memcmp(yh.data(), ref.data(), ref.size() * 8) == 0; st_same = st_same && same;
                } else {                                          // fresh encryptions: other ciphertexts of the same images -- bench.py decrypts these
                    vector<uint64_t> yh = y.toHost();
                    ofstream o(outputs + ".streamed", ios::binary); o.write((const char *)yh.data(), (streamsize)((size_t)distinct * 10 * ctw * 8));
                }
            }
            allgather(0, 0, 0, 0);
            const auto ts = chrono::high_resolution_clock::now();
            const int total_launches = stream_steps * launches;
            upload(0, true);
            for (int l = 0; l < total_launches; l++) {
                if (l + 1 < total_launches) upload((l + 1) & 1, l >= 1);        // the next launch's bytes travel while this one is evaluated
                run(l & 1);
            }
            if (crc_stream_sync(context, compute) < 0) throw runtime_error("crc_stream_sync");
            st_dt = chrono::duration<double>(chrono::high_resolution_clock::now() - ts).count();
            st_images = (double)total_launches * launch; st_bytes = st_images * (double)unit * 8;
            allgather((uint64_t)(st_dt * 1e9), (uint64_t)st_images, 0, 0);
            double m = 0.0, im = 0.0;
            for (int r = 0; r < world; r++) { m = max(m, (double)gathered[(size_t)r * 4] * 1e-9); im += (double)gathered[(size_t)r * 4 + 1]; }
            st_dt = m; st_images = im; st_bytes *= world;
            for (int i = 0; i < 2; i++) { crc_event_destroy(context, copied[i]); crc_event_destroy(context, consumed[i]); }
            crc_host_free(context, pinned);
            char buf[1400];
            snprintf(buf, sizeof buf,
                "%s{\"mode\": %s, \"images_per_s\": %.4f, \"elapsed_s\": %.6f, \"steps\": %d, \"h2d_GBps\": %.2f, \"bytes_per_image\": %zu, "
                     "\"outputs_identical_to_resident\": %s, \"how\": \"two device input buffers; the next launch's images are copied from page-locked host "
                         "memory on a copy stream "
                     "while the compute stream evaluates the current one (events order the two)%s\"}", streamed_json.empty() ? "" : ", ", jstr(mode).c_str(),
                         st_images / st_dt, st_dt,
                     stream_steps, st_bytes / st_dt / 1e9, (size_t)(unit * 8), pt ? "null" : same ? "true" : "false",
                     pt ? "; the 784 pixel plaintexts per image are encrypted on the device (crc_encrypt_dev_forms, NTT-form result) in front of the first "
                         "layer" : "");
            streamed_json += buf;
        }

        printf("{\"host\": \"C++ host classes (crcnn_amd/host: CnnBuilder / Network::fuse / Network::forward) over the C ABI\", \"model\": %s, \"n\": %d, "
            "\"k\": %d, \"batch\": %d, \"chunk\": %d, "
               "\"group\": %d, \"steps\": %d, \"warmup\": %d, \"rank\": %d, \"world\": %d, \"images_per_s\": %.4f, \"elapsed_s\": %.6f, \"ms_per_step\": "
                   "%.3f, \"ms_per_image\": %.4f, \"setup_s\": %.1f, ",
               jstr(model).c_str(), n, k, launches * launch, head, group, steps, warmup, rank, world, images_all / dt_max, dt_max, dt_max / steps * 1e3,
                   dt / images * 1e3, setup_s);
        printf("\"last_timed_launch_identical_to_first\": %s, ", timed_same ? "true" : "false");
        printf("\"timing\": \"all-gather (barrier) + stream synchronisation on both sides of the timed steps on every rank; elapsed_s = max over ranks, "
            "images_per_s = all ranks' images / that\", "
               "\"per_rank\": [");
        for (int r = 0; r < world; r++)
            printf("%s{\"rank\": %d, \"elapsed_s\": %.6f, \"images\": %llu, \"free_after_first_launch\": %llu, \"parameters\": %llu}", r ? ", " : "", r,
                (double)per_rank[(size_t)r * 4] * 1e-9,
                   (unsigned long long)per_rank[(size_t)r * 4 + 1], (unsigned long long)per_rank[(size_t)r * 4 + 2],
                       (unsigned long long)per_rank[(size_t)r * 4 + 3]);
        printf("], ");
        if (comm) printf("\"weight_broadcast\": {\"weights_via\": %s, \"via\": \"Network::broadcastParameters (crc_broadcast_weights: %s; per-rank checksums "
            "compared with the root's)\", \"bytes\": %zu, "
                         "\"seconds\": %.3f, \"GBps\": %.2f, \"seconds_include\": \"the root's lift + NTT of its plaintext weights, the per-rank checksums "
                             "and their all-gather\", "
                         "\"host_threads_per_rank\": %d}, ", jstr(weights_via == "floats" ? "floats: every rank encodes + transforms the weights itself, only "
                             "the evaluation keys are sent" : "broadcast: the root's encoded weights and the evaluation keys are sent").c_str(),
                             getenv("CRC_COMM_TRANSPORT") &&
                             !strcmp(getenv("CRC_COMM_TRANSPORT"), "shm") ?
                         "shared-memory rehearsal transport" : "ncclBroadcast over RCCL in <= 1 GiB pieces", bcast_bytes, bcast_s, bcast_s > 0 ?
                             bcast_bytes / bcast_s / 1e9 : 0.0, crc_host_thread_limit());
        else printf("\"weight_broadcast\": null, ");
        if (!streamed_json.empty()) printf("\"streamed\": [%s], ", streamed_json.c_str());
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
        printf("], \"layer_before_reenc\": %d, \"T_REENC_ms_per_image\": %.4f, \"matrix_cores\": %s", net.layer_before_reenc, t_reenc / images, matrix_cores ? "true" :
            "false");
        if (sync_each) printf(", \"ms_per_image_sync_each\": %.4f", t_sync / images);
        printf(", \"layer_timing\": \"HIP events on the launch stream around every Layer::forward inside the timed region\", ");
        printf("\"hbm\": {\"total_bytes\": %zu, \"free_before_build\": %zu, \"free_after_first_launch\": %zu, \"parameters\": %zu, \"activation_slots\": %zu, "
            "\"work_buffer\": %zu, "
               "\"evaluation_keys\": %zu, \"input_launch\": %zu}}\n", total, free0, free1, plan.parameters, plan.activations, plan.work, plan.keys,
                   (size_t)launch * imgw * 8);
        fflush(stdout);
        if (comm) crc_comm_destroy(comm);
        setStream(nullptr);
        delParameters();
        if (!st_same) fprintf(stderr, "bench_host: a streamed launch's outputs differ from the resident launch's\n");
        if (!timed_same) fprintf(stderr, "bench_host: the last timed launch's outputs differ from the first launch's\n");
        return st_same && timed_same ? 0 : 4;
    } catch (const exception &e) {
        fprintf(stderr, "bench_host: %s\n", e.what());
        fflush(stderr);
        // (no unwinding of device state after a failed launch or allocation: the message above is the report)
        _exit(2);
    }
}
