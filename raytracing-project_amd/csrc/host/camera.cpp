// camera.cpp - camera.sav files and the corner rays of a frame.
//
// The reference takes its camera from the OpenGL state: modelview = the
// trackball matrix (traqueboule.h:52-55), projection =
// gluPerspective(50, w/h, 0.01, 10) (main.cpp:209), viewport (0,0,w,h), and
// un-projects the window points of the four corner pixels at depth 0 and 1
// (main.cpp:52-55, 532-552).  Pixel (x_I, y_I) maps to window (x_I, h - y_I).
// Here the same mapping is evaluated in double precision without OpenGL: a
// window point goes to normalised device coordinates, then through the inverse
// of projection*modelview, then the homogeneous divide.
#include <cmath>
#include <cstdio>

#include "rt_host.hpp"

namespace rtamd {

bool readCameraSav(const char* path, double modelview[16], std::vector<Vec3Df>& lights)
{
    FILE* f = fopen(path, "r");
    if (!f) return false;
    bool ok = true;
    for (int i = 0; i < 16 && ok; ++i) {
        float v;
        ok = fscanf(f, "%f ", &v) == 1;
        modelview[i] = v; // the reference reads into a float first (main.cpp:286-287)
    }
    int n = 0;
    ok = ok && fscanf(f, "%d", &n) == 1 && n >= 0 && n < 100000;
    lights.clear();
    for (int i = 0; i < n && ok; ++i) {
        Vec3Df l;
        ok = fscanf(f, "%f %f %f", &l[0], &l[1], &l[2]) == 3;
        if (ok) lights.push_back(l);
    }
    fclose(f);
    return ok;
}

bool writeCameraSav(const char* path, const double modelview[16], const std::vector<Vec3Df>& lights)
{
    FILE* f = fopen(path, "wb");
    if (!f) return false;
    for (int i = 0; i < 16; ++i) fprintf(f, "%f ", modelview[i]);
    fprintf(f, "\r\n%d\r\n", (int)lights.size());
    for (const Vec3Df& l : lights) fprintf(f, "%f %f %f\r\n", l[0], l[1], l[2]);
    fclose(f);
    return true;
}

namespace {

// column-major 4x4 helpers: m[col*4 + row]
void mul44(const double a[16], const double b[16], double out[16]) // out = a * b
{
    for (int c = 0; c < 4; ++c)
        for (int r = 0; r < 4; ++r) {
            double s = 0;
            for (int k = 0; k < 4; ++k) s += a[k * 4 + r] * b[c * 4 + k];
            out[c * 4 + r] = s;
        }
}

bool invert44(const double in[16], double out[16]) // Gauss-Jordan with partial pivoting
{
    double a[4][8];
    for (int r = 0; r < 4; ++r)
        for (int c = 0; c < 4; ++c) {
            a[r][c] = in[c * 4 + r];
            a[r][4 + c] = r == c ? 1.0 : 0.0;
        }
    for (int col = 0; col < 4; ++col) {
        int piv = col;
        for (int r = col + 1; r < 4; ++r)
            if (std::fabs(a[r][col]) > std::fabs(a[piv][col])) piv = r;
        if (a[piv][col] == 0.0) return false;
        if (piv != col)
            for (int c = 0; c < 8; ++c) std::swap(a[piv][c], a[col][c]);
        const double inv = 1.0 / a[col][col];
        for (int c = 0; c < 8; ++c) a[col][c] *= inv;
        for (int r = 0; r < 4; ++r) {
            if (r == col) continue;
            const double f = a[r][col];
            if (f != 0.0)
                for (int c = 0; c < 8; ++c) a[r][c] -= f * a[col][c];
        }
    }
    for (int r = 0; r < 4; ++r)
        for (int c = 0; c < 4; ++c) out[c * 4 + r] = a[r][4 + c];
    return true;
}

} // namespace

void cameraCorners(const double modelview[16], int width, int height, float corners[24])
{
    // gluPerspective(50, (float)w/h, 0.01, 10)
    const double fovy = 50.0, zn = 0.01, zf = 10.0;
    const double aspect = (double)((float)width / height);
    const double half = fovy / 2.0 * 3.14159265358979323846 / 180.0;
    const double cot = std::cos(half) / std::sin(half);
    double       proj[16] = {0};
    proj[0] = cot / aspect;
    proj[5] = cot;
    proj[10] = -(zf + zn) / (zf - zn);
    proj[11] = -1.0;
    proj[14] = -2.0 * zn * zf / (zf - zn);
    double pm[16], inv[16];
    mul44(proj, modelview, pm);
    const bool ok = invert44(pm, inv);
    const int  px[4][2] = {{0, 0}, {0, height - 1}, {width - 1, 0}, {width - 1, height - 1}};
    for (int c = 0; c < 4; ++c)
        for (int far = 0; far < 2; ++far) {
            const double wx = px[c][0], wy = height - px[c][1], wz = far;
            const double ndc[4] = {wx / width * 2.0 - 1.0, wy / height * 2.0 - 1.0, wz * 2.0 - 1.0, 1.0};
            double       o[4] = {0, 0, 0, 1};
            if (ok)
                for (int r = 0; r < 4; ++r)
                    o[r] = inv[r] * ndc[0] + inv[4 + r] * ndc[1] + inv[8 + r] * ndc[2] + inv[12 + r] * ndc[3];
            float* dst = corners + (2 * c + far) * 3;
            for (int a = 0; a < 3; ++a) dst[a] = (ok && o[3] != 0.0) ? (float)(o[a] / o[3]) : 0.0f;
        }
}

} // namespace rtamd
