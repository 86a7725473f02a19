// rt_render - headless front end of the 'r' render path.  The reference starts
// a GLUT window and renders when 'r' is pressed (main.cpp:254-260) after 'S'
// has restored camera.sav (main.cpp:281-307); this program does the same two
// steps from the command line and writes the same four files.
//
// usage: rt_render <scene.obj> <camera.sav> <width> <height> [msaa=1] [out_dir=.] [device=0] [focus_x focus_y]
// focus_x focus_y: key 'f' with the mouse over that pixel before the render (raytracing.cpp:504-507)
#include <cstdio>
#include <cstdlib>

#include "rt_host.hpp"

int main(int argc, char** argv)
{
    if (argc < 5) {
        fprintf(stderr, "usage: %s <scene.obj> <camera.sav> <width> <height> [msaa] [out_dir] [device] [focus_x focus_y]\n",
                argv[0]);
        return 2;
    }
    const int   w = atoi(argv[3]), h = atoi(argv[4]);
    const int   msaa = argc > 5 ? atoi(argv[5]) : 1;
    const char* out = argc > 6 ? argv[6] : ".";
    const int   device = argc > 7 ? atoi(argv[7]) : 0;
    if (!rtamd::init(argv[1], device)) {
        fprintf(stderr, "init failed: %s\n", rtamd::lastError().c_str());
        return 1;
    }
    double modelview[16];
    if (!rtamd::readCameraSav(argv[2], modelview, rtamd::MyLightPositions)) {
        fprintf(stderr, "cannot read %s\n", argv[2]);
        return 1;
    }
    rtamd::MySettings.msaa = msaa;
    rtamd::RayTracer tracer(modelview, w, h);
    if (argc > 9 && !tracer.setFocusAtPixel(atof(argv[8]), atof(argv[9]))) {
        fprintf(stderr, "focus pick failed: %s\n", rtamd::lastError().c_str());
        return 1;
    }
    const double     secs = tracer.doDaRayTracingShizz(out);
    if (secs < 0) {
        fprintf(stderr, "render failed: %s\n", rtamd::lastError().c_str());
        return 1;
    }
    const RtStats& s = tracer.stats;
    printf("%.3f s wall, kernels %.3f ms, %llu closest + %llu shadow rays\n", secs, s.kernel_ms,
           (unsigned long long)s.rays_closest, (unsigned long long)s.rays_shadow);
    rtamd::shutdown();
    return 0;
}
