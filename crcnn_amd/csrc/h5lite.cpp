// h5lite.cpp -- minimal built-in HDF5 reader for the model files the reference loads.
//
// Replaces LoadH5::getDataVfloat (CrCNN/src/H5Easy.cpp:584-644) as used by CnnBuilder::getPretrained
// (cnnBuilder.cpp:20-23): "give me dataset <name> as a flat float vector".  The reference links libhdf5 for this; the
// files written by PlainModel/ToH5.py are the simplest kind of HDF5 (superblock v0, one root group with a v1 B-tree +
// local heap, v1 object headers, contiguous little-endian IEEE float32 datasets), so we parse exactly that subset from
// the published HDF5 file-format specification and fail loudly (CRC_ERR_IO) on anything else.
#include "../../include/crcnn_hip.h"
#include <cstdint>
#include <cstdio>
#include <cstring>
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

extern "C" int crc_h5_dataset_count(const char *path, const char *name, size_t *count)
{
    if (!path || !name || !count) return CRC_ERR_INVALID_ARGUMENT;
    Reader r; if (!r.open(path)) return CRC_ERR_IO;
    const Dataset *d = r.find(name); if (!d) return CRC_ERR_NOT_FOUND;
    Info i = r.info(d->header); if (!i.ok) return CRC_ERR_IO;
    *count = (size_t)i.count;
    return CRC_OK;
}
extern "C" int crc_h5_read_f32(const char *path, const char *name, float *out, size_t cap, size_t *count)
{
    if (!path || !name || !out) return CRC_ERR_INVALID_ARGUMENT;
    Reader r; if (!r.open(path)) return CRC_ERR_IO;
    const Dataset *d = r.find(name); if (!d) return CRC_ERR_NOT_FOUND;
    Info i = r.info(d->header); if (!i.ok) return CRC_ERR_IO;
    if (count) *count = (size_t)i.count;
    if (i.count > cap) return CRC_ERR_INVALID_ARGUMENT;
    std::memcpy(out, &r.f.d[i.data_off], (size_t)i.data_len);      // x86-64 host is little endian, like the file
    return CRC_OK;
}
extern "C" int crc_h5_list(const char *path, char *names, size_t cap)
{
    if (!path || !names || cap == 0) return CRC_ERR_INVALID_ARGUMENT;
    Reader r; if (!r.open(path)) return CRC_ERR_IO;
    std::string all;
    for (auto &s : r.sets) { all += s.name; all += '\n'; }
    if (all.size() + 1 > cap) return CRC_ERR_INVALID_ARGUMENT;
    std::memcpy(names, all.c_str(), all.size() + 1);
    return (int)r.sets.size();
}
