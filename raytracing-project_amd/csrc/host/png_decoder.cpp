// png_decoder.cpp - PNG to RGBA8, what the reference obtains from the Corona library
// (mesh.h:117-135: corona::OpenImage(path, PF_R8G8B8A8), rows top first).
//
// Written against the PNG specification (ISO/IEC 15948): all five colour types, bit
// depths 1-16, the five scanline filters, Adam7 interlacing, PLTE and tRNS; ancillary chunks
// other than tRNS are skipped, chunk CRCs are verified.  zlib does the inflate.  Samples
// deeper than 8 bits keep their high byte, shallower ones are scaled to 0..255; images
// without transparency get alpha 255.  PNG is lossless, so the bytes are those of any
// conforming decoder (checked in the tests against Pillow and against the pre-decoded
// textures the scenes ship with).
#include "rt_host.hpp"

#include <zlib.h>

#include <cstdio>
#include <cstring>

namespace rtamd {
namespace {

uint32_t be32(const uint8_t* p) { return ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | p[3]; }

int paeth(int a, int b, int c)
{
    const int p = a + b - c, pa = abs(p - a), pb = abs(p - b), pc = abs(p - c);
    return (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c);
}

// Undo the filter of one scanline in place.  prev = the line above (nullptr for the first).
bool unfilter(int type, uint8_t* cur, const uint8_t* prev, size_t n, size_t bpp)
{
    switch (type) {
    case 0: return true;
    case 1:
        for (size_t i = bpp; i < n; ++i) cur[i] = (uint8_t)(cur[i] + cur[i - bpp]);
        return true;
    case 2:
        if (prev)
            for (size_t i = 0; i < n; ++i) cur[i] = (uint8_t)(cur[i] + prev[i]);
        return true;
    case 3:
        for (size_t i = 0; i < n; ++i) {
            const int a = i >= bpp ? cur[i - bpp] : 0, b = prev ? prev[i] : 0;
            cur[i] = (uint8_t)(cur[i] + ((a + b) >> 1));
        }
        return true;
    case 4:
        for (size_t i = 0; i < n; ++i) {
            const int a = i >= bpp ? cur[i - bpp] : 0, b = prev ? prev[i] : 0, c = (prev && i >= bpp) ? prev[i - bpp] : 0;
            cur[i] = (uint8_t)(cur[i] + paeth(a, b, c));
        }
        return true;
    default: return false;
    }
}

struct Header {
    uint32_t w = 0, h = 0;
    int      depth = 0, ctype = 0, interlace = 0;
    int      channels() const { return ctype == 0 ? 1 : ctype == 2 ? 3 : ctype == 3 ? 1 : ctype == 4 ? 2 : 4; }
};

// sample s (0-based) of a scanline with `depth` bits per sample
unsigned sample_at(const uint8_t* line, size_t s, int depth)
{
    switch (depth) {
    case 8: return line[s];
    case 16: return ((unsigned)line[2 * s] << 8) | line[2 * s + 1];
    default: {
        const size_t bit = s * (size_t)depth;
        const int    shift = 8 - depth - (int)(bit & 7);
        return (line[bit >> 3] >> shift) & ((1u << depth) - 1u);
    }
    }
}

} // namespace

bool decodePng(const uint8_t* data, size_t size, Texture& out, std::string& err)
{
    static const uint8_t sig[8] = {0x89, 'P', 'N', 'G', 0x0d, 0x0a, 0x1a, 0x0a};
    if (size < 8 || memcmp(data, sig, 8)) return err = "not a PNG file", false;
    Header               hd;
    std::vector<uint8_t> idat, plte, trns;
    bool                 have_ihdr = false, have_iend = false, have_trns = false;
    size_t               pos = 8;
    while (pos + 12 <= size && !have_iend) {
        const uint32_t len = be32(data + pos);
        const uint8_t* type = data + pos + 4;
        if (len > size - pos - 12) return err = "truncated chunk", false;
        const uint8_t* body = data + pos + 8;
        if ((uint32_t)crc32(crc32(0L, type, 4), body, len) != be32(body + len)) return err = "chunk CRC mismatch", false;
        if (!memcmp(type, "IHDR", 4)) {
            if (len != 13 || have_ihdr) return err = "bad IHDR", false;
            hd.w = be32(body), hd.h = be32(body + 4);
            hd.depth = body[8], hd.ctype = body[9], hd.interlace = body[12];
            if (body[10] != 0 || body[11] != 0) return err = "unknown compression or filter method", false;
            have_ihdr = true;
        } else if (!have_ihdr) {
            return err = "IHDR is not the first chunk", false;
        } else if (!memcmp(type, "PLTE", 4)) {
            plte.assign(body, body + len);
        } else if (!memcmp(type, "tRNS", 4)) {
            trns.assign(body, body + len);
            have_trns = true;
        } else if (!memcmp(type, "IDAT", 4)) {
            idat.insert(idat.end(), body, body + len);
        } else if (!memcmp(type, "IEND", 4)) {
            have_iend = true;
        } else if (!(type[0] & 0x20)) {
            return err = "unknown critical chunk", false;
        }
        pos += 12 + (size_t)len;
    }
    if (!have_ihdr || !have_iend) return err = "missing IHDR or IEND", false;
    const int d = hd.depth, ct = hd.ctype;
    const bool depth_ok = (ct == 0 && (d == 1 || d == 2 || d == 4 || d == 8 || d == 16)) ||
                          (ct == 3 && (d == 1 || d == 2 || d == 4 || d == 8)) ||
                          ((ct == 2 || ct == 4 || ct == 6) && (d == 8 || d == 16));
    if (!depth_ok) return err = "invalid colour type / bit depth", false;
    if (hd.w == 0 || hd.h == 0 || hd.w > 16384 || hd.h > 16384) return err = "image size out of range", false;
    if (hd.interlace > 1) return err = "unknown interlace method", false;
    if (ct == 3 && (plte.empty() || plte.size() % 3)) return err = "palette image without a valid PLTE", false;

    const int    ch = hd.channels();
    const size_t bits_pp = (size_t)ch * d, bpp = bits_pp >= 8 ? bits_pp / 8 : 1;
    auto         line_bytes = [&](size_t w) { return (w * bits_pp + 7) / 8; };
    // passes: (x0, y0, dx, dy); one pass for a non-interlaced image
    static const int adam7[7][4] = {{0, 0, 8, 8}, {4, 0, 8, 8}, {0, 4, 4, 8}, {2, 0, 4, 4}, {0, 2, 2, 4}, {1, 0, 2, 2}, {0, 1, 1, 2}};
    static const int whole[1][4] = {{0, 0, 1, 1}};
    const int(*pass)[4] = hd.interlace ? adam7 : whole;
    const int n_pass = hd.interlace ? 7 : 1;
    size_t    raw_size = 0;
    for (int p = 0; p < n_pass; ++p) {
        const size_t pw = (hd.w - pass[p][0] + pass[p][2] - 1) / pass[p][2], ph = (hd.h - pass[p][1] + pass[p][3] - 1) / pass[p][3];
        if (hd.w > (uint32_t)pass[p][0] && hd.h > (uint32_t)pass[p][1] && pw && ph) raw_size += ph * (1 + line_bytes(pw));
    }
    std::vector<uint8_t> raw(raw_size);
    {
        z_stream zs;
        memset(&zs, 0, sizeof zs);
        if (inflateInit(&zs) != Z_OK) return err = "zlib init failed", false;
        zs.next_in = idat.data(), zs.avail_in = (uInt)idat.size();
        zs.next_out = raw.data(), zs.avail_out = (uInt)raw.size();
        const int rc = inflate(&zs, Z_FINISH);
        const size_t got = raw.size() - zs.avail_out;
        inflateEnd(&zs);
        if ((rc != Z_STREAM_END && rc != Z_OK && rc != Z_BUF_ERROR) || got != raw.size())
            return err = "image data does not inflate to the expected size", false;
    }
    out.width = (int)hd.w, out.height = (int)hd.h;
    out.rgba.assign((size_t)hd.w * hd.h * 4, 0);
    const unsigned maxv = (1u << d) - 1u;
    // colour key of tRNS for grey / RGB images (full-depth sample values)
    unsigned key[3] = {0, 0, 0};
    const bool keyed = have_trns && ((ct == 0 && trns.size() >= 2) || (ct == 2 && trns.size() >= 6));
    if (keyed)
        for (int c = 0; c < (ct == 0 ? 1 : 3); ++c) key[c] = ((unsigned)trns[2 * c] << 8) | trns[2 * c + 1];
    auto to8 = [&](unsigned v) -> uint8_t {
        if (d == 8) return (uint8_t)v;
        if (d == 16) return (uint8_t)(v >> 8);
        return (uint8_t)((v * 255u + maxv / 2) / maxv);
    };
    size_t off = 0;
    for (int p = 0; p < n_pass; ++p) {
        if (hd.w <= (uint32_t)pass[p][0] || hd.h <= (uint32_t)pass[p][1]) continue;
        const size_t pw = (hd.w - pass[p][0] + pass[p][2] - 1) / pass[p][2], ph = (hd.h - pass[p][1] + pass[p][3] - 1) / pass[p][3];
        const size_t lb = line_bytes(pw);
        const uint8_t* prev = nullptr;
        for (size_t y = 0; y < ph; ++y) {
            uint8_t* line = raw.data() + off + 1;
            if (!unfilter(raw[off], line, prev, lb, bpp)) return err = "unknown filter type", false;
            const size_t oy = (size_t)pass[p][1] + y * pass[p][3];
            for (size_t x = 0; x < pw; ++x) {
                const size_t ox = (size_t)pass[p][0] + x * pass[p][2];
                uint8_t*     px = &out.rgba[4 * (oy * hd.w + ox)];
                switch (ct) {
                case 0: {
                    const unsigned v = sample_at(line, x, d);
                    px[0] = px[1] = px[2] = to8(v);
                    px[3] = (keyed && v == key[0]) ? 0 : 255;
                    break;
                }
                case 2: {
                    const unsigned r = sample_at(line, 3 * x, d), g = sample_at(line, 3 * x + 1, d), b = sample_at(line, 3 * x + 2, d);
                    px[0] = to8(r), px[1] = to8(g), px[2] = to8(b);
                    px[3] = (keyed && r == key[0] && g == key[1] && b == key[2]) ? 0 : 255;
                    break;
                }
                case 3: {
                    const unsigned i = sample_at(line, x, d);
                    if (3 * (size_t)i + 2 >= plte.size()) return err = "palette index out of range", false;
                    px[0] = plte[3 * i], px[1] = plte[3 * i + 1], px[2] = plte[3 * i + 2];
                    px[3] = (have_trns && i < trns.size()) ? trns[i] : 255;
                    break;
                }
                case 4: {
                    px[0] = px[1] = px[2] = to8(sample_at(line, 2 * x, d));
                    px[3] = to8(sample_at(line, 2 * x + 1, d));
                    break;
                }
                default: {
                    for (int c = 0; c < 4; ++c) px[c] = to8(sample_at(line, 4 * x + c, d));
                    break;
                }
                }
            }
            prev = line;
            off += 1 + lb;
        }
    }
    return true;
}

bool readPng(const std::string& path, Texture& t, std::string& err)
{
    FILE* f = fopen(path.c_str(), "rb");
    if (!f) return err = "cannot open " + path, false;
    std::vector<uint8_t> buf;
    uint8_t              chunk[65536];
    size_t               n;
    while ((n = fread(chunk, 1, sizeof chunk, f)) > 0) buf.insert(buf.end(), chunk, chunk + n);
    fclose(f);
    if (!decodePng(buf.data(), buf.size(), t, err)) return false;
    t.path = path;
    return true;
}

} // namespace rtamd

extern "C" {
// Test hook: decodes a PNG held in memory.  rgba (may be null to query the size) receives
// width * height * 4 bytes.  Returns 0, or -1 with the reason in rth_last_error().
int rth_decode_png(const uint8_t* data, size_t size, int* width, int* height, uint8_t* rgba, size_t rgba_capacity)
{
    rtamd::Texture t;
    std::string    err;
    if (!data || !rtamd::decodePng(data, size, t, err)) {
        rtamd::setLastError(err.empty() ? "null data" : err);
        return -1;
    }
    if (width) *width = t.width;
    if (height) *height = t.height;
    if (rgba) {
        if (rgba_capacity < t.rgba.size()) {
            rtamd::setLastError("output buffer too small");
            return -1;
        }
        memcpy(rgba, t.rgba.data(), t.rgba.size());
    }
    return 0;
}
}
