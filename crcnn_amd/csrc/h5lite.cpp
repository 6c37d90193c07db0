// h5lite.cpp -- minimal built-in HDF5 reader for the model files the reference loads.
//
// Replaces LoadH5::getDataVfloat (CrCNN/src/H5Easy.cpp:584-644) as used by CnnBuilder::getPretrained
// (cnnBuilder.cpp:20-23): "give me dataset <name> as a flat float vector".  The reference links libhdf5 for this; the
// files written by PlainModel/ToH5.py are the simplest kind of HDF5 (superblock v0, one root group with a v1 B-tree +
// local heap, v1 object headers, contiguous little-endian IEEE float32 datasets), so we parse exactly that subset from
// the published HDF5 file-format specification.  Anything else -- a model re-exported with h5py(libver="latest"), chunked or compressed datasets -- goes to libhdf5
// itself, exactly as the reference reads it, when the library is on the machine (dlopen at first need: the engine does not link it); without it such a file fails
// loudly (CRC_ERR_IO).  CRC_H5_BACKEND=lite|hdf5 forces one reader (tests).
#include "../../include/crcnn_hip.h"
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <dlfcn.h>
#include <mutex>
#include <string>
#include <vector>

namespace {
struct File {
    std::vector<uint8_t> d;
    bool load(const char *path)
    {
        FILE *f = std::fopen(path, "rb");
        if (!f) return false;
        std::fseek(f, 0, SEEK_END); long sz = std::ftell(f); std::fseek(f, 0, SEEK_SET);
        if (sz <= 0) { std::fclose(f); return false; }
        d.resize((size_t)sz);
        bool ok = std::fread(d.data(), 1, (size_t)sz, f) == (size_t)sz;
        std::fclose(f);
        return ok;
    }
    bool in(uint64_t off, uint64_t len) const { return off <= d.size() && len <= d.size() - off; }
    // base-relative address from the file -> absolute offset, or ~0 when the sum wraps / leaves the file (every caller then fails its `in` check)
    uint64_t rel(uint64_t base, uint64_t a) const { const uint64_t s = base + a; return (s < base || s > d.size()) ? ~(uint64_t)0 : s; }
    uint64_t u(uint64_t off, int bytes) const { uint64_t v = 0; for (int i = 0; i < bytes; i++) v |= (uint64_t)d[off + i] << (8 * i); return v; }
};

struct Dataset { std::string name; uint64_t header = 0; };
struct Info { uint64_t count = 0, data_off = 0, data_len = 0; bool ok = false; };

struct Reader {
    File f; uint64_t base = 0; int so = 8, sl = 8;
    std::vector<Dataset> sets;

    bool open(const char *path)
    {
        if (!f.load(path)) return false;
        static const uint8_t sig[8] = {0x89, 'H', 'D', 'F', '\r', '\n', 0x1a, '\n'};
        if (!f.in(0, 96) || std::memcmp(f.d.data(), sig, 8)) return false;
        const int ver = f.d[8];
        if (ver > 1) return false;                                   // superblock v2/v3 (newer libver) not needed here
        so = f.d[13]; sl = f.d[14];
        if (so != 8 || sl != 8) return false;
        uint64_t p = 24 + (ver == 1 ? 4 : 0);
        if (!f.in(p, 4 * (uint64_t)so)) return false;
        base = f.u(p, so); p += 4 * (uint64_t)so;                    // base, free-space, eof, driver-info addresses
        if (base > f.d.size()) return false;
        // root group symbol-table entry: name offset, header address, cache type, reserved, scratch (btree, heap)
        if (!f.in(p, 2 * (uint64_t)so + 8 + 16)) return false;
        const uint32_t cache = (uint32_t)f.u(p + 2 * so, 4);
        if (cache != 1) return false;
        const uint64_t bt = f.u(p + 2 * so + 8, so), heap = f.u(p + 2 * so + 8 + so, so);
        const uint64_t heap_abs = f.rel(base, heap);
        if (!f.in(heap_abs, 8 + 2 * (uint64_t)sl + so) || std::memcmp(&f.d[heap_abs], "HEAP", 4)) return false;
        const uint64_t heap_data = f.rel(base, f.u(heap_abs + 8 + 2 * sl, so));
        if (!f.in(heap_data, 1)) return false;
        return walk(f.rel(base, bt), heap_data, 0);
    }
    bool walk(uint64_t node, uint64_t heap_data, int depth)
    {
        if (depth > 16 || !f.in(node, 8 + 2 * (uint64_t)so) || std::memcmp(&f.d[node], "TREE", 4) || f.d[node + 4] != 0) return false;
        const int level = f.d[node + 5]; const int used = (int)f.u(node + 6, 2);
        uint64_t p = node + 8 + 2 * so;
        for (int e = 0; e < used; e++) {
            p += sl;                                                 // key
            if (!f.in(p, so)) return false;
            const uint64_t child = f.rel(base, f.u(p, so)); p += so;
            if (level > 0) { if (!walk(child, heap_data, depth + 1)) return false; continue; }
            if (!f.in(child, 8) || std::memcmp(&f.d[child], "SNOD", 4)) return false;
            const int nsym = (int)f.u(child + 6, 2);
            if (sets.size() + (size_t)nsym > 65536) return false;    // (a model file has a few dozen datasets)
            uint64_t q = child + 8;
            for (int s = 0; s < nsym; s++, q += 2 * so + 24) {
                if (!f.in(q, 2 * (uint64_t)so + 24)) return false;
                const uint64_t noff = f.rel(heap_data, f.u(q, so));
                if (!f.in(noff, 1)) return false;
                Dataset ds; ds.header = f.rel(base, f.u(q + so, so));
                for (uint64_t c = noff; c < f.d.size() && f.d[c]; c++) ds.name.push_back((char)f.d[c]);
                sets.push_back(ds);
            }
        }
        return true;
    }
    // parse a v1 object header: float32 LE contiguous dataset -> element count + data extent
    Info info(uint64_t oh) const
    {
        Info r;
        if (!f.in(oh, 16) || f.d[oh] != 1) return r;
        int nmsg = (int)f.u(oh + 2, 2);
        uint64_t p = oh + 16, end = p + f.u(oh + 8, 4);
        if (!f.in(p, end - p)) return r;
        bool have_space = false, have_type = false, have_layout = false;
        std::vector<std::pair<uint64_t, uint64_t>> cont;
        for (int m = 0; m < nmsg; m++) {
            while (p + 8 > end) {
                if (cont.empty()) return r;
                p = cont.back().first; const uint64_t len = cont.back().second; cont.pop_back();
                if (!f.in(p, len)) return r;
                end = p + len;
            }
            if (!f.in(p, 8)) return r;
            const int type = (int)f.u(p, 2); const uint64_t sz = f.u(p + 2, 2); const uint64_t b = p + 8;
            if (!f.in(b, sz) || b + sz > end) return r;              // the body must lie inside this header block: every read below is checked against sz
            if (type == 0x0001) {                                    // dataspace
                if (sz < 8) return r;
                const int ver = f.d[b], rank = f.d[b + 1];
                if (ver != 1 && ver != 2) return r;
                const uint64_t hdr = ver == 1 ? 8 : 4;
                if (rank > 32 || sz < hdr + (uint64_t)rank * sl) return r;
                const uint64_t dp = b + hdr;
                r.count = 1;
                for (int i = 0; i < rank; i++) {
                    const uint64_t dim = f.u(dp + (uint64_t)i * sl, sl);
                    if (dim != 0 && r.count > (((uint64_t)1 << 40) / dim)) return r;       // more elements than any file holds: overflow guard for count and count*4
                    r.count *= dim;
                }
                have_space = true;
            } else if (type == 0x0003) {                             // datatype: class 1 (floating point), 4 bytes, little endian
                if (sz < 8) return r;
                const int cls = f.d[b] & 0x0f; const uint32_t size = (uint32_t)f.u(b + 4, 4);
                if (cls != 1 || size != 4 || (f.d[b + 1] & 1)) return r;
                have_type = true;
            } else if (type == 0x0008) {                             // data layout v3, contiguous
                if (sz < 2 + (uint64_t)so + sl || f.d[b] != 3 || f.d[b + 1] != 1) return r;
                r.data_off = f.rel(base, f.u(b + 2, so)); r.data_len = f.u(b + 2 + so, sl);
                have_layout = true;
            } else if (type == 0x0010) {                             // object header continuation
                if (sz < (uint64_t)so + sl || cont.size() > 64) return r;
                cont.push_back({f.rel(base, f.u(b, so)), f.u(b + so, sl)});
            } else if (type == 0x000B) return r;                     // filter pipeline (compression): not supported
            p = b + sz;
        }
        r.ok = have_space && have_type && have_layout && r.data_len == r.count * 4 && f.in(r.data_off, r.data_len);
        return r;
    }
    const Dataset *find(const char *name) const
    {
        std::string n(name); if (!n.empty() && n[0] == '/') n = n.substr(1);
        for (auto &s : sets) if (s.name == n) return &s;
        return nullptr;
    }
};
}  // namespace

// ---- libhdf5 behind dlopen (H5Easy.cpp:584-644 calls the same library through its C++ wrapper: H5File::openDataSet, DataSet::read(..., PredType::NATIVE_FLOAT)) -----------
namespace {
typedef int64_t hid;                                       // hid_t since HDF5 1.10 (older libraries, where it is an int, are not used)
struct Hdf5 {
    void *lib = nullptr;
    int (*H5open)() = nullptr; int (*H5get_libversion)(unsigned *, unsigned *, unsigned *) = nullptr;
    hid (*H5Fopen)(const char *, unsigned, hid) = nullptr; int (*H5Fclose)(hid) = nullptr;
    hid (*H5Dopen2)(hid, const char *, hid) = nullptr; int (*H5Dclose)(hid) = nullptr; hid (*H5Dget_space)(hid) = nullptr; hid (*H5Dget_type)(hid) = nullptr;
    long long (*H5Sget_simple_extent_npoints)(hid) = nullptr; int (*H5Sclose)(hid) = nullptr;
    int (*H5Tget_class)(hid) = nullptr; int (*H5Tclose)(hid) = nullptr;
    int (*H5Dread)(hid, hid, hid, hid, hid, void *) = nullptr;
    int (*H5Literate)(hid, int, int, unsigned long long *, int (*)(hid, const char *, const void *, void *), void *) = nullptr;
    int (*H5Eset_auto2)(hid, void *, void *) = nullptr;
    hid native_float = -1;
    bool ok = false;
    Hdf5()
    {
        const char *env = std::getenv("CRC_LIBHDF5");
        const char *names[] = {env, "libhdf5.so", "libhdf5_serial.so", "libhdf5.so.310", "libhdf5.so.200", "libhdf5.so.103", "libhdf5_serial.so.103", "/opt/conda/lib/libhdf5.so"};
        for (const char *nm : names) if (nm && *nm && (lib = dlopen(nm, RTLD_NOW | RTLD_LOCAL))) break;
        if (!lib) return;
        auto sym = [&](const char *n) { return dlsym(lib, n); };
#define LOADSYM(f) *(void **)(&f) = sym(#f)
        LOADSYM(H5open); LOADSYM(H5get_libversion); LOADSYM(H5Fopen); LOADSYM(H5Fclose); LOADSYM(H5Dopen2); LOADSYM(H5Dclose); LOADSYM(H5Dget_space); LOADSYM(H5Dget_type);
        LOADSYM(H5Sget_simple_extent_npoints); LOADSYM(H5Sclose); LOADSYM(H5Tget_class); LOADSYM(H5Tclose); LOADSYM(H5Dread); LOADSYM(H5Eset_auto2);
#undef LOADSYM
        *(void **)(&H5Literate) = sym("H5Literate");
        if (!H5Literate) *(void **)(&H5Literate) = sym("H5Literate1");          // 1.12+: the versioned names (the callback's first two arguments, all it uses, are the same)
        if (!H5open || !H5get_libversion || !H5Fopen || !H5Fclose || !H5Dopen2 || !H5Dclose || !H5Dget_space || !H5Dget_type || !H5Sget_simple_extent_npoints || !H5Sclose ||
            !H5Tget_class || !H5Tclose || !H5Dread || !H5Literate) return;
        unsigned maj = 0, min = 0, rel = 0;
        if (H5open() < 0 || H5get_libversion(&maj, &min, &rel) < 0 || maj != 1 || min < 10) return;       // 64-bit hid_t
        const hid *nf = (const hid *)sym("H5T_NATIVE_FLOAT_g");                  // what the H5T_NATIVE_FLOAT macro reads (valid after H5open)
        if (!nf || *nf < 0) return;
        native_float = *nf;
        if (H5Eset_auto2) H5Eset_auto2(0 /* H5E_DEFAULT */, nullptr, nullptr);   // failures are return codes here, not a stack trace on stderr
        ok = true;
    }
};
Hdf5 &hdf5() { static Hdf5 h; return h; }
std::mutex g_h5_mu;                                        // the library is not thread-safe unless it was built so
enum { kBackendAuto, kBackendLite, kBackendHdf5 };
int backend() { const char *e = std::getenv("CRC_H5_BACKEND"); return !e ? kBackendAuto : !std::strcmp(e, "lite") ? kBackendLite : !std::strcmp(e, "hdf5") ? kBackendHdf5 : kBackendAuto; }

// count only (out == nullptr) or read: CRC_OK, CRC_ERR_NOT_FOUND (no such dataset), CRC_ERR_IO (no library / not HDF5 / not a float dataset), CRC_ERR_INVALID_ARGUMENT (cap)
int hdf5_read(const char *path, const char *name, float *out, size_t cap, size_t *count)
{
    Hdf5 &h = hdf5();
    if (!h.ok) return CRC_ERR_IO;
    std::lock_guard<std::mutex> lk(g_h5_mu);
    unsigned long long fsize = 0;
    { FILE *fp = std::fopen(path, "rb"); if (!fp) return CRC_ERR_IO; std::fseek(fp, 0, SEEK_END); const long sz = std::ftell(fp); std::fclose(fp); if (sz <= 0) return CRC_ERR_IO; fsize = (unsigned long long)sz; }
    const hid f = h.H5Fopen(path, 0 /* H5F_ACC_RDONLY */, 0 /* H5P_DEFAULT */);
    if (f < 0) return CRC_ERR_IO;
    int rc = CRC_OK;
    const hid d = h.H5Dopen2(f, name, 0);
    if (d < 0) rc = CRC_ERR_NOT_FOUND;
    else {
        const hid sp = h.H5Dget_space(d), ty = h.H5Dget_type(d);
        const long long np = sp >= 0 ? h.H5Sget_simple_extent_npoints(sp) : -1;
        // (the reference reads PredType::NATIVE_FLOAT from float datasets only; an element count no file of this size can hold -- 4096:1 is beyond any filter --
        // is a damaged header, not a dataset to allocate for)
        if (np < 0 || ty < 0 || h.H5Tget_class(ty) != 1 /* H5T_FLOAT */ || (unsigned long long)np > fsize * 1024ULL) rc = CRC_ERR_IO;
        else {
            if (count) *count = (size_t)np;
            if (out) { if ((size_t)np > cap) rc = CRC_ERR_INVALID_ARGUMENT; else if (np && h.H5Dread(d, h.native_float, 0 /* H5S_ALL */, 0, 0, out) < 0) rc = CRC_ERR_IO; }
        }
        if (ty >= 0) h.H5Tclose(ty);
        if (sp >= 0) h.H5Sclose(sp);
        h.H5Dclose(d);
    }
    h.H5Fclose(f);
    return rc;
}
int list_cb(hid, const char *name, const void *, void *data) { std::string *all = (std::string *)data; *all += name; *all += '\n'; return 0; }
int hdf5_list(const char *path, std::string &all, int &n)
{
    Hdf5 &h = hdf5();
    if (!h.ok) return CRC_ERR_IO;
    std::lock_guard<std::mutex> lk(g_h5_mu);
    const hid f = h.H5Fopen(path, 0, 0);
    if (f < 0) return CRC_ERR_IO;
    unsigned long long idx = 0;
    const int rc = h.H5Literate(f, 0 /* H5_INDEX_NAME */, 0 /* H5_ITER_INC */, &idx, list_cb, &all);
    h.H5Fclose(f);
    if (rc < 0) return CRC_ERR_IO;
    n = 0; for (char c : all) n += c == '\n';
    return CRC_OK;
}
}  // namespace

extern "C" int crc_h5_backend_available(void) { return hdf5().ok ? 1 : 0; }

extern "C" int crc_h5_dataset_count(const char *path, const char *name, size_t *count)
{
    if (!path || !name || !count) return CRC_ERR_INVALID_ARGUMENT;
    const int be = backend();
    if (be != kBackendHdf5) {
        Reader r;
        if (r.open(path)) {
            const Dataset *d = r.find(name); if (!d) return CRC_ERR_NOT_FOUND;
            Info i = r.info(d->header);
            if (i.ok) { *count = (size_t)i.count; return CRC_OK; }
        }
        if (be == kBackendLite) return CRC_ERR_IO;
    }
    return hdf5_read(path, name, nullptr, 0, count);          // (a layout the built-in reader does not parse: libhdf5 when the machine has it)
}
extern "C" int crc_h5_read_f32(const char *path, const char *name, float *out, size_t cap, size_t *count)
{
    if (!path || !name || !out) return CRC_ERR_INVALID_ARGUMENT;
    const int be = backend();
    if (be != kBackendHdf5) {
        Reader r;
        if (r.open(path)) {
            const Dataset *d = r.find(name); if (!d) return CRC_ERR_NOT_FOUND;
            Info i = r.info(d->header);
            if (i.ok) {
                if (count) *count = (size_t)i.count;
                if (i.count > cap) return CRC_ERR_INVALID_ARGUMENT;
                std::memcpy(out, &r.f.d[i.data_off], (size_t)i.data_len);      // x86-64 host is little endian, like the file
                return CRC_OK;
            }
        }
        if (be == kBackendLite) return CRC_ERR_IO;
    }
    return hdf5_read(path, name, out, cap, count);
}
extern "C" int crc_h5_list(const char *path, char *names, size_t cap)
{
    if (!path || !names || cap == 0) return CRC_ERR_INVALID_ARGUMENT;
    const int be = backend();
    std::string all; int n = -1;
    if (be != kBackendHdf5) {
        Reader r;
        if (r.open(path)) { for (auto &s : r.sets) { all += s.name; all += '\n'; } n = (int)r.sets.size(); }
        else if (be == kBackendLite) return CRC_ERR_IO;
    }
    if (n < 0) { const int rc = hdf5_list(path, all, n); if (rc) return rc; }
    if (all.size() + 1 > cap) return CRC_ERR_INVALID_ARGUMENT;
    std::memcpy(names, all.c_str(), all.size() + 1);
    return n;
}
