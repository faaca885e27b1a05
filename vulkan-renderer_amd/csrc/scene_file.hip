// Scene cache: the on-disk form of the arrays the culling path consumes (Engine.hpp:215-218:
// vertices, indices, meshlets, meshletData) plus the per-instance meshlet ranges, so that the
// reference's load-time pipeline (loadModel / optimizeGeometry / createMeshlets, Engine.cpp:243-628)
// runs once per asset instead of once per start.  Host code only; little-endian; every section
// starts on a 16-byte boundary so a mapped file can be handed to the upload calls as it is.
//
//   0   char[8]  "MCULLSC\1"
//   8   u32 version (1), u32 header bytes (96)
//   16  u64 vertex_count, index_count, meshlet_data_words, meshlet_count, range_count
//   56  u64 payload bytes, u64 FNV-1a-64 of the payload, 24 reserved bytes (zero)
//   96  payload: mcull_vertex[], u32 indices[], u32 meshletData[], mcull_meshlet[], u32 range_offset[],
//       u32 range_count[]  (each padded with zeros to a multiple of 16 bytes)
#include <cstdio>
#include <cstring>
#include <new>
#include <vector>

#include "../../include/mcull.h"

namespace {

constexpr char kMagic[8] = {'M', 'C', 'U', 'L', 'L', 'S', 'C', 1};
constexpr uint32_t kVersion = 1, kHeaderBytes = 96;

struct Header {
    char magic[8];
    uint32_t version, header_bytes;
    uint64_t vertex_count, index_count, meshlet_data_words, meshlet_count, range_count;
    uint64_t payload_bytes, checksum, reserved[3];
};
static_assert(sizeof(Header) == kHeaderBytes, "scene file header layout");

uint64_t pad16(uint64_t n) { return (n + 15u) & ~15ull; }

uint64_t fnv1a64(uint64_t h, const void *data, uint64_t n) {
    const unsigned char *p = static_cast<const unsigned char *>(data);
    for (uint64_t i = 0; i < n; i++) {
        h ^= p[i];
        h *= 0x100000001b3ull;
    }
    return h;
}

struct Section {
    const void *data;
    uint64_t bytes;
};

uint64_t payload_bytes(const Header &h) {
    return pad16(h.vertex_count * sizeof(mcull_vertex)) + pad16(h.index_count * 4) + pad16(h.meshlet_data_words * 4) +
           pad16(h.meshlet_count * sizeof(mcull_meshlet)) + 2 * pad16(h.range_count * 4);
}

// Counts that cannot describe a real file (overflowing sizes) are rejected before any allocation.
bool sane(const Header &h) {
    const uint64_t lim = 1ull << 40;
    return h.vertex_count < lim && h.index_count < lim && h.meshlet_data_words < lim && h.meshlet_count < lim &&
           h.range_count < lim;
}

// The header is believed only when the file really is as long as it says: a header that promises more than
// the file holds must not size anybody's allocation.  Leaves the stream just behind the header.
int read_header(FILE *f, Header *h) {
    if (std::fread(h, 1, sizeof *h, f) != sizeof *h) return MCULL_E_IO;
    if (std::memcmp(h->magic, kMagic, 8) != 0 || h->version != kVersion || h->header_bytes != kHeaderBytes || !sane(*h) ||
        h->payload_bytes != payload_bytes(*h))
        return MCULL_E_INVALID;
    if (std::fseek(f, 0, SEEK_END) != 0) return MCULL_E_IO;
    const long long size = std::ftell(f);
    if (size < 0 || std::fseek(f, (long)kHeaderBytes, SEEK_SET) != 0) return MCULL_E_IO;
    if ((unsigned long long)size != h->payload_bytes + kHeaderBytes) return MCULL_E_INVALID;
    return MCULL_OK;
}

bool same_counts(const Header &h, const mcull_scene_counts &c) {
    return h.vertex_count == c.vertex_count && h.index_count == c.index_count && h.meshlet_data_words == c.meshlet_data_words &&
           h.meshlet_count == c.meshlet_count && h.range_count == c.range_count && h.checksum == c.checksum;
}

}  // namespace

extern "C" {

int mcull_scene_save(const char *path, const mcull_vertex *vertices, uint64_t vertex_count, const uint32_t *indices,
                     uint64_t index_count, const uint32_t *meshlet_data, uint64_t meshlet_data_words,
                     const mcull_meshlet *meshlets, uint64_t meshlet_count, const uint32_t *range_offset,
                     const uint32_t *range_count, uint64_t n_ranges) {
    if (!path) return MCULL_E_INVALID;
    if ((!vertices && vertex_count) || (!indices && index_count) || (!meshlet_data && meshlet_data_words) ||
        (!meshlets && meshlet_count) || ((!range_offset || !range_count) && n_ranges))
        return MCULL_E_INVALID;
    Header h;
    std::memset(&h, 0, sizeof h);
    std::memcpy(h.magic, kMagic, 8);
    h.version = kVersion;
    h.header_bytes = kHeaderBytes;
    h.vertex_count = vertex_count;
    h.index_count = index_count;
    h.meshlet_data_words = meshlet_data_words;
    h.meshlet_count = meshlet_count;
    h.range_count = n_ranges;
    if (!sane(h)) return MCULL_E_INVALID;
    h.payload_bytes = payload_bytes(h);
    const Section sec[6] = {{vertices, vertex_count * sizeof(mcull_vertex)}, {indices, index_count * 4},
                            {meshlet_data, meshlet_data_words * 4},           {meshlets, meshlet_count * sizeof(mcull_meshlet)},
                            {range_offset, n_ranges * 4},                     {range_count, n_ranges * 4}};
    static const unsigned char zeros[16] = {0};
    uint64_t sum = 0xcbf29ce484222325ull;
    for (const Section &s : sec) {
        sum = fnv1a64(sum, s.data, s.bytes);
        sum = fnv1a64(sum, zeros, pad16(s.bytes) - s.bytes);
    }
    h.checksum = sum;
    FILE *f = std::fopen(path, "wb");
    if (!f) return MCULL_E_IO;
    bool ok = std::fwrite(&h, 1, sizeof h, f) == sizeof h;
    for (const Section &s : sec) {
        ok = ok && (s.bytes == 0 || std::fwrite(s.data, 1, s.bytes, f) == s.bytes);
        const uint64_t pad = pad16(s.bytes) - s.bytes;
        ok = ok && (pad == 0 || std::fwrite(zeros, 1, pad, f) == pad);
    }
    ok = (std::fclose(f) == 0) && ok;
    if (!ok) {
        std::remove(path);  // never leave a truncated cache behind
        return MCULL_E_IO;
    }
    return MCULL_OK;
}

int mcull_scene_info(const char *path, mcull_scene_counts *out) {
    if (!path || !out) return MCULL_E_INVALID;
    FILE *f = std::fopen(path, "rb");
    if (!f) return MCULL_E_IO;
    Header h;
    const int rc = read_header(f, &h);
    std::fclose(f);
    if (rc) return rc;
    out->vertex_count = h.vertex_count;
    out->index_count = h.index_count;
    out->meshlet_data_words = h.meshlet_data_words;
    out->meshlet_count = h.meshlet_count;
    out->range_count = h.range_count;
    out->checksum = h.checksum;
    return MCULL_OK;
}

int mcull_scene_load(const char *path, const mcull_scene_counts *expected, mcull_vertex *vertices, uint32_t *indices,
                     uint32_t *meshlet_data, mcull_meshlet *meshlets, uint32_t *range_offset, uint32_t *range_count) {
    if (!path || !expected) return MCULL_E_INVALID;
    FILE *f = std::fopen(path, "rb");
    if (!f) return MCULL_E_IO;
    Header h;
    int rc = read_header(f, &h);
    // the caller sized its arrays from `expected`: a file that has changed since then must not be read into them
    if (!rc && !same_counts(h, *expected)) rc = MCULL_E_INVALID;
    if (rc) {
        std::fclose(f);
        return rc;
    }
    struct Dst {
        void *data;
        uint64_t bytes;
    } dst[6] = {{vertices, h.vertex_count * sizeof(mcull_vertex)}, {indices, h.index_count * 4},
                {meshlet_data, h.meshlet_data_words * 4},           {meshlets, h.meshlet_count * sizeof(mcull_meshlet)},
                {range_offset, h.range_count * 4},                  {range_count, h.range_count * 4}};
    uint64_t sum = 0xcbf29ce484222325ull;
    std::vector<unsigned char> skip;
    for (const Dst &d : dst) {
        // a section the caller does not want (NULL) is still read: the checksum covers the whole payload
        void *to = d.data;
        if (!to && d.bytes) {
            try {
                skip.resize(d.bytes);
            } catch (const std::bad_alloc &) {
                std::fclose(f);
                return MCULL_E_NOMEM;
            }
            to = skip.data();
        }
        unsigned char pad[16];
        const uint64_t npad = pad16(d.bytes) - d.bytes;
        if ((d.bytes && std::fread(to, 1, d.bytes, f) != d.bytes) || (npad && std::fread(pad, 1, npad, f) != npad)) {
            std::fclose(f);
            return MCULL_E_IO;
        }
        sum = fnv1a64(sum, to, d.bytes);
        sum = fnv1a64(sum, pad, npad);
    }
    std::fclose(f);
    return sum == h.checksum ? MCULL_OK : MCULL_E_INVALID;
}

int mcull_scene_upload(mcull_ctx *ctx, const char *path) {
    if (!ctx || !path) return MCULL_E_INVALID;
    mcull_scene_counts info;
    int rc = mcull_scene_info(path, &info);
    if (rc) return rc;
    std::vector<mcull_vertex> v;
    std::vector<uint32_t> md, ro, rcount;
    std::vector<mcull_meshlet> ml;
    try {  // nothing may throw across the C boundary
        v.resize(info.vertex_count);
        md.resize(info.meshlet_data_words);
        ro.resize(info.range_count);
        rcount.resize(info.range_count);
        ml.resize(info.meshlet_count);
    } catch (const std::bad_alloc &) {
        return MCULL_E_NOMEM;
    }
    rc = mcull_scene_load(path, &info, v.data(), nullptr, md.data(), ml.data(), ro.data(), rcount.data());
    if (rc) return rc;
    if ((rc = mcull_upload_meshlets(ctx, ml.data(), ml.size(), MCULL_MEM_HOST))) return rc;
    if (!v.empty() && (rc = mcull_upload_geometry(ctx, v.data(), v.size(), md.data(), md.size(), MCULL_MEM_HOST))) return rc;
    return mcull_set_instance_ranges(ctx, info.range_count ? ro.data() : nullptr, info.range_count ? rcount.data() : nullptr,
                                     info.range_count);
}

}  // extern "C"
