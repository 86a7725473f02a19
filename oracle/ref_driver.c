/*
 * ref_driver.c - loads _ref/libref.so (the reference sources + ref_harness.cpp)
 * with lazy symbol binding and runs its probe.  Lazy binding matters: the
 * reference's Mesh::loadMesh calls into the Corona image library, which does
 * not exist here; the probe never calls loadMesh, so that symbol is never
 * looked up.  TEST INFRASTRUCTURE, build container only.
 */
#include <dlfcn.h>
#include <libgen.h>
#include <limits.h>
#include <stdio.h>
#include <string.h>
#include <unistd.h>

int main(int argc, char **argv)
{
    char exe[PATH_MAX], lib[PATH_MAX + 16];
    ssize_t n = readlink("/proc/self/exe", exe, sizeof exe - 1);
    if (n <= 0) return 1;
    exe[n] = 0;
    snprintf(lib, sizeof lib, "%s/libref.so", dirname(exe));
    void *h = dlopen(lib, RTLD_LAZY | RTLD_LOCAL);
    if (!h) {
        fprintf(stderr, "%s\n", dlerror());
        return 1;
    }
    int (*ref_main)(int, char **) = (int (*)(int, char **))dlsym(h, "ref_main");
    if (!ref_main) return 1;
    return ref_main(argc, argv);
}
