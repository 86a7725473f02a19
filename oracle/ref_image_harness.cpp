/*
 * ref_image_harness.cpp - drives the REAL imageWriter.h of the reference
 * (included where it lies under /root/reference by oracle/Makefile, target
 * "ref") and dumps what its Image class computes and writes, so that the
 * image half of the CPU restatement in rt_oracle.c (depth normalisation, the
 * 13x13 depth-of-field kernel, the blur, the BMP/PPM writers) can be pinned
 * against it.  TEST INFRASTRUCTURE, build container only: the committed
 * tests/golden/image_*.npz vectors are made from this program's output by
 * tools/make_image_golden.py.
 *
 * What runs from the reference, unmodified: RGBValue (imageWriter.h:23-44),
 * Image::setPixel (:99), getDepthRange / normailzeDepthRange (:113-135),
 * gauseanMap (:137-181), blurrLine / blurrAllLines (:210-237), writeImagePPM
 * (:253), writeImageBMP (:283), writeDepthBMP (:342), writeBlurredBMP (:406)
 * and the per-row writeImageBMP(filename, x, y, w, h) (:472), called in the
 * order of the 'r' path (main.cpp:325, 404-407, 371-377) on one Image object,
 * which matters: the writers share the member buffer imageC.
 *
 * usage: refimage <in.bin> <outdir>
 *   in.bin : int32 W, H; float focus; float rgb[W*H*3]; float depth[W*H]
 *   outdir : progress.bmp result.ppm result.bmp depth.bmp blurred.bmp dump.bin
 *   dump.bin: float image[W*H*3] (after the RGBValue clamp), depth1[W*H]
 *             (after writeDepthBMP), depth2[W*H] (after writeBlurredBMP),
 *             blurr[(W-12)*(H-12)*3], then 7 x 169 floats: gauseanMap of the
 *             reference's own list .001 .01 .1 .3 .7 .9 1.0 (main.cpp:238-245)
 */
#include "config.h"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <iostream>
#include <string>
#include <vector>

#include "imageWriter.h"

float focusDepth = 0; // raytracing.cpp:17 defines it in the full program

static void rd(FILE *f, void *p, size_t n)
{
    if (fread(p, 1, n, f) != n) {
        fprintf(stderr, "refimage: short read\n");
        exit(2);
    }
}

int main(int argc, char **argv)
{
    if (argc < 3) return 1;
    FILE *in = fopen(argv[1], "rb");
    if (!in) return 1;
    const std::string dir = argv[2];
    int   W, H;
    float focus;
    rd(in, &W, 4), rd(in, &H, 4), rd(in, &focus, 4);
    std::vector<float> rgb((size_t)3 * W * H), depth((size_t)W * H);
    rd(in, rgb.data(), rgb.size() * 4);
    rd(in, depth.data(), depth.size() * 4);
    fclose(in);
    focusDepth = focus;

    Image img(W, H);
    const std::string progress = dir + "/progress.bmp";
    img.writeImageBMP(progress.c_str()); // main.cpp:325, the blank file the row updates seek into
    for (int y = 0; y < H; ++y) {
        for (int x = 0; x < W; ++x) {
            const float *p = &rgb[3 * ((size_t)y * W + x)];
            img.setPixel(x, y, RGBValue(p[0], p[1], p[2]), depth[(size_t)y * W + x]); // main.cpp:404
        }
        img.writeImageBMP(progress.c_str(), 0, y, img._width, 1); // main.cpp:407
    }
    FILE *out = fopen((dir + "/dump.bin").c_str(), "wb");
    if (!out) return 1;
    fwrite(img._image.data(), 4, img._image.size(), out);
    img.writeImagePPM((dir + "/result.ppm").c_str()); // main.cpp:371-377
    img.writeImageBMP((dir + "/result.bmp").c_str());
    img.writeDepthBMP((dir + "/depth.bmp").c_str());
    fwrite(img._depth.data(), 4, img._depth.size(), out);
    img.writeBlurredBMP((dir + "/blurred.bmp").c_str());
    fwrite(img._depth.data(), 4, img._depth.size(), out);
    fwrite(img._blurr.data(), 4, img._blurr.size(), out);
    const float list[7] = {0.001f, 0.01f, 0.1f, 0.3f, 0.7f, 0.9f, 1.0f}; // main.cpp:239-245
    for (int k = 0; k < 7; ++k) {
        std::vector<std::vector<float>> map = img.gauseanMap(list[k]);
        for (int i = 0; i < 13; ++i) fwrite(map[i].data(), 4, 13, out);
    }
    fclose(out);
    return 0;
}
