// rt_render - headless front end of the 'r' render path.  The reference starts
// a GLUT window and renders when 'r' is pressed (main.cpp:254-260) after 'S'
// has restored camera.sav (main.cpp:281-307); this program does the same two
// steps from the command line and writes the same four files.
//
// usage: rt_render <scene.obj> <camera.sav> <width> <height> [msaa=1] [out_dir=.] [device=0] [focus_x focus_y]
//        rt_render --gpus N <scene.obj> <camera.sav> <width> <height> [msaa=1] [out_dir=.]
// focus_x focus_y: key 'f' with the mouse over that pixel before the render (raytracing.cpp:504-507)
// --gpus N: the frame is shared by N processes, one per GPU (devices 0..N-1), started here
// before anything touches a GPU; they replace the reference's thread bands (main.cpp:329-339).
// Rank 0 makes the RCCL unique id and hands it to the others through pipes, every rank renders
// its tiles, rank 0 gathers them, post-processes and writes the files.
#include <signal.h>
#include <sys/wait.h>
#include <unistd.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "rt_host.hpp"

static int renderOne(int argc, char** argv, void* comm, int rank, int n_ranks, int device)
{
    const int   w = atoi(argv[3]), h = atoi(argv[4]);
    const int   msaa = argc > 5 ? atoi(argv[5]) : 1;
    const char* out = argc > 6 ? argv[6] : ".";
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
    if (n_ranks == 1 && argc > 9 && !tracer.setFocusAtPixel(atof(argv[8]), atof(argv[9]))) {
        fprintf(stderr, "focus pick failed: %s\n", rtamd::lastError().c_str());
        return 1;
    }
    tracer.shareFrame(comm, rank, n_ranks);
    const double secs = tracer.doDaRayTracingShizz(out);
    if (secs < 0) {
        fprintf(stderr, "render failed: %s\n", rtamd::lastError().c_str());
        return 1;
    }
    const RtStats& s = tracer.stats;
    printf("rank %d of %d: %.3f s wall, kernels %.3f ms, %llu closest + %llu shadow rays\n", rank, n_ranks, secs,
           s.kernel_ms, (unsigned long long)s.rays_closest, (unsigned long long)s.rays_shadow);
    rtamd::shutdown();
    return 0;
}

static int usage(const char* exe)
{
    fprintf(stderr,
            "usage: %s <scene.obj> <camera.sav> <width> <height> [msaa] [out_dir] [device] [focus_x focus_y]\n"
            "       %s --gpus N <scene.obj> <camera.sav> <width> <height> [msaa] [out_dir]\n",
            exe, exe);
    return 2;
}

int main(int argc, char** argv)
{
    if (argc > 2 && !strcmp(argv[1], "--gpus")) {
        const int n = atoi(argv[2]);
        if (n < 1 || n > 64 || argc < 7) return usage(argv[0]);
        std::vector<char*> rest{argv[0]};
        for (int i = 3; i < argc; ++i) rest.push_back(argv[i]);
        // one pipe per rank > 0 for the unique id; all made before the first fork
        std::vector<int> rd((size_t)n, -1), wr((size_t)n, -1);
        for (int r = 1; r < n; ++r) {
            int fd[2];
            if (pipe(fd) != 0) {
                perror("pipe");
                return 1;
            }
            rd[r] = fd[0], wr[r] = fd[1];
        }
        std::vector<pid_t> kids;
        for (int r = 0; r < n; ++r) {
            const pid_t pid = fork(); // this process has not touched a GPU
            if (pid < 0) {
                perror("fork");
                return 1;
            }
            if (pid == 0) {
                // keep only the ends this rank uses: a reader must see end-of-file when rank 0 dies before
                // it has written the id, not wait on write ends its siblings still hold
                for (int k = 1; k < n; ++k) {
                    if (r != 0 && wr[k] >= 0) close(wr[k]);
                    if (k != r && rd[k] >= 0) close(rd[k]);
                }
                unsigned char id[RT_RCCL_ID_BYTES];
                void*         comm = nullptr;
                if (n > 1) {
                    if (r == 0) {
                        if (rt_rccl_unique_id(id) != RT_OK) {
                            fprintf(stderr, "rank 0: %s\n", rt_last_error());
                            _exit(1);
                        }
                        for (int k = 1; k < n; ++k)
                            if (write(wr[k], id, sizeof id) != (ssize_t)sizeof id) _exit(1);
                    } else if (read(rd[r], id, sizeof id) != (ssize_t)sizeof id) {
                        fprintf(stderr, "rank %d: no unique id from rank 0\n", r);
                        _exit(1);
                    }
                    if (rt_rccl_comm_create(id, n, r, r, &comm) != RT_OK) {
                        fprintf(stderr, "rank %d: %s\n", r, rt_last_error());
                        _exit(1);
                    }
                }
                const int rc = renderOne((int)rest.size(), rest.data(), comm, r, n, r);
                if (comm) rt_rccl_comm_destroy(comm);
                fflush(stdout);
                _exit(rc);
            }
            kids.push_back(pid);
        }
        for (int r = 1; r < n; ++r) close(rd[r]), close(wr[r]);
        // A rank that fails leaves its peers waiting for it (in the communicator's set-up or in the
        // gather): the first failure ends the others, so the program returns instead of hanging.
        int    worst = 0;
        bool   stopping = false;
        size_t left = kids.size();
        while (left > 0) {
            int         st = 0;
            const pid_t done = waitpid(-1, &st, 0);
            if (done < 0) break;
            size_t at = kids.size();
            for (size_t i = 0; i < kids.size(); ++i)
                if (kids[i] == done) at = i;
            if (at == kids.size()) continue;
            kids[at] = -1, --left;
            const int rc = WIFEXITED(st) ? WEXITSTATUS(st) : 1;
            if (rc > worst) worst = rc;
            if (rc != 0 && left > 0 && !stopping) {
                stopping = true;
                fprintf(stderr, "rank %zu failed (status %d): stopping the other %zu\n", at, rc, left);
                for (pid_t k : kids)
                    if (k > 0) kill(k, SIGTERM);
                for (int tries = 0; tries < 50 && left > 0; ++tries) { // up to 5 s to go by themselves
                    pid_t w;
                    while (left > 0 && (w = waitpid(-1, &st, WNOHANG)) > 0)
                        for (pid_t& k : kids)
                            if (k == w) k = -1, --left;
                    if (left > 0) usleep(100000);
                }
                for (pid_t k : kids)
                    if (k > 0) kill(k, SIGKILL);
            }
        }
        return worst;
    }
    if (argc < 5) return usage(argv[0]);
    return renderOne(argc, argv, nullptr, 0, 1, argc > 7 ? atoi(argv[7]) : 0);
}
