// image_io.cpp - the four result files of the 'r' path (main.cpp:371-377),
// byte-compatible with the reference's writers in imageWriter.h:
//   * 8-bit value = (unsigned char)(v * 255.0f), truncating (imageWriter.h:270);
//     NaN and values outside the int range become 0, which is what the x86
//     float->int conversion followed by the narrowing gives;
//   * BMP: 14 + 40 byte headers, 24 bpp, height stored negative (top-down),
//     BGR byte order; the size field holds 3*w*h for result.bmp/depth.bmp and
//     w*h for blurred.bmp, not the file size (imageWriter.h:298-301, 426-430);
//     rows are laid out with the 4-byte-aligned stride but only 3*w*h bytes are
//     written (imageWriter.h:316-331), so widths that are not a multiple of 4
//     come out skewed exactly as the reference's do;
//   * depth.bmp: grey = 1 - normalised depth.  The reference's inner loop runs
//     over 3*w "pixels" per row (imageWriter.h:379-389) and so scribbles over
//     the two rows that follow, but every row rewrites its own bytes and its
//     padding afterwards and the index guard keeps the last rows inside the
//     buffer: what is left in the end is plain grey rows plus padding bytes of
//     value 1, which is what is written here directly;
//   * the three BMP writers fill one byte buffer that belongs to the Image
//     (imageWriter.h:239) and blurred.bmp's padding loop writes three bytes
//     early (imageWriter.h:456-458), so for widths whose rows need padding the
//     last three bytes of every blurred.bmp row are whatever the previous
//     writer (depth.bmp on the 'r' path) left there.  Image::imageC keeps that
//     buffer between calls for the same reason;
//   * the per-row update (imageWriter.h:472-488) seeks to 54 + (Y*w + x)*3,
//     i.e. rows packed without the stride, into an existing file.
#include <cstdio>
#include <cstring>

#include "rt_host.hpp"

namespace rtamd {

namespace {

inline unsigned char quantize(float v)
{
    const float x = v * 255.0f;
    if (!(x == x) || x >= 2147483648.0f || x < -2147483648.0f) return 0;
    return (unsigned char)(int)x;
}

void putHeaders(FILE* f, unsigned sizeField, int w, int h)
{
    unsigned char fh[14] = {'B', 'M', 0, 0, 0, 0, 0, 0, 0, 0, 54, 0, 0, 0};
    unsigned char ih[40] = {40, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 1, 0, 24, 0};
    const unsigned uw = (unsigned)w, uh = (unsigned)(-h);
    for (int i = 0; i < 4; ++i) {
        fh[2 + i] = (unsigned char)(sizeField >> (8 * i));
        ih[4 + i] = (unsigned char)(uw >> (8 * i));
        ih[8 + i] = (unsigned char)(uh >> (8 * i));
    }
    fwrite(fh, 1, sizeof fh, f);
    fwrite(ih, 1, sizeof ih, f);
}

inline int stride(int w) { return ((24 * w + 31) / 32) * 4; }

// The 8-bit form of a float plane: the given bytes if the GPU already made them, else quantised here.
const unsigned char* bytesOf(const unsigned char* ready, const std::vector<float>& plane,
                             std::vector<unsigned char>& tmp)
{
    if (ready) return ready;
    tmp.resize(plane.size());
    for (size_t i = 0; i < plane.size(); ++i) tmp[i] = quantize(plane[i]);
    return tmp.data();
}

// rgb: w*h*3 bytes in RGB order.  padShift: how many bytes early the padding loop writes
// (imageWriter.h:326-328 writes in place, :456-458 three bytes early).
bool writeColour(const char* filename, const unsigned char* rgb, int w, int h, unsigned sizeField, int padShift,
                 std::vector<unsigned char>& bytes)
{
    FILE* f = fopen(filename, "wb");
    if (!f) {
        fprintf(stderr, "dump file problem... file\n");
        return false;
    }
    putHeaders(f, sizeField, w, h);
    const int rs = stride(w);
    if (bytes.size() < (size_t)rs * h + 8) bytes.resize((size_t)rs * h + 8, 0);
    for (int y = 0; y < h; ++y) {
        unsigned char*       row = bytes.data() + (size_t)y * rs;
        const unsigned char* src = rgb + (size_t)y * w * 3;
        for (int x = 0; x < w; ++x) {
            row[3 * x] = src[3 * x + 2];
            row[3 * x + 1] = src[3 * x + 1];
            row[3 * x + 2] = src[3 * x];
        }
        for (int i = 3 * w; i < rs; ++i) row[i - padShift] = 1;
    }
    const bool ok = fwrite(bytes.data(), (size_t)w * h * 3, 1, f) == 1;
    if (!ok) fprintf(stderr, "Dump file problem... fwrite\n");
    fclose(f);
    return ok;
}

} // namespace

bool Image::writeImagePPM(const char* filename) const
{
    FILE* f = fopen(filename, "wb");
    if (!f) {
        fprintf(stderr, "dump file problem... file\n");
        return false;
    }
    fprintf(f, "P6\n%i %i\n255\n", width, height);
    std::vector<unsigned char> tmp;
    const bool ok = fwrite(bytesOf(rgb8, rgb, tmp), (size_t)width * height * 3, 1, f) == 1;
    fclose(f);
    return ok;
}

bool Image::writeImageBMP(const char* filename) const
{
    std::vector<unsigned char> tmp;
    return writeColour(filename, bytesOf(rgb8, rgb, tmp), width, height, (unsigned)(3 * width * height), 0, imageC);
}

// imageWriter.h:472-488: rows y..y+h-1, columns x..x+w-1 into the file writeImageBMP made
bool Image::writeImageBMP(const char* filename, int x, int y, int w, int h) const
{
    FILE* f = fopen(filename, "r+b");
    if (!f) return true; // the reference's fstream fails silently and still reports success
    std::vector<unsigned char> row((size_t)w * 3);
    for (int Y = y; Y < y + h; ++Y) {
        for (int j = x; j < x + w; ++j) {
            const size_t   p = ((size_t)Y * width + j) * 3;
            unsigned char* d = &row[(size_t)(j - x) * 3];
            if (rgb8) d[0] = rgb8[p + 2], d[1] = rgb8[p + 1], d[2] = rgb8[p];
            else d[0] = quantize(rgb[p + 2]), d[1] = quantize(rgb[p + 1]), d[2] = quantize(rgb[p]);
        }
        fseek(f, 54 + ((long)Y * w + x) * 3, SEEK_SET);
        fwrite(row.data(), 1, row.size(), f);
    }
    fclose(f);
    return true;
}

bool Image::writeBlurredBMP(const char* filename) const
{
    const int bw = width - 12, bh = height - 12;
    if (bw <= 0 || bh <= 0 || (!blurred8 && blurred.size() != (size_t)bw * bh * 3)) return false;
    if (imageC.size() < (size_t)stride(width) * height + 8) imageC.resize((size_t)stride(width) * height + 8, 0);
    std::vector<unsigned char> tmp;
    return writeColour(filename, bytesOf(blurred8, blurred, tmp), bw, bh, (unsigned)(bw * bh), 3, imageC);
}

bool Image::writeDepthBMP(const char* filename) const
{
    FILE* f = fopen(filename, "wb");
    if (!f) {
        fprintf(stderr, "dump file problem... file\n");
        return false;
    }
    putHeaders(f, (unsigned)(3 * width * height), width, height);
    const int                   rs = stride(width);
    const size_t                npx = (size_t)width * height;
    std::vector<unsigned char>& bytes = imageC;
    if (bytes.size() < (size_t)rs * height + 8) bytes.resize((size_t)rs * height + 8, 0);
    for (int y = 0; y < height; ++y) {
        unsigned char* row = bytes.data() + (size_t)y * rs;
        for (int x = 0; x < width; ++x) {
            const size_t        p = (size_t)y * width + x;
            const unsigned char v = depth8 ? depth8[p] : quantize(1 - depth[p]);
            row[3 * x] = row[3 * x + 1] = row[3 * x + 2] = v;
        }
        for (int i = 3 * width; i < rs; ++i) row[i] = 1;
    }
    const bool ok = fwrite(bytes.data(), npx * 3, 1, f) == 1;
    fclose(f);
    return ok;
}

} // namespace rtamd
