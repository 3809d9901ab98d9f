// yaha -- the command line (reference Main.c).  Everything lives in libyaha_hip.so (yaha_main); this file only decides how the process ends: a finished run
// flushes its output and leaves through _exit, so that device memory, page-locked buffers and the index mapping are released by the kernel in one go instead of
// buffer by buffer (YAHA_FAST_EXIT tells the library not to tear its contexts down first; set YAHA_KEEP_TEARDOWN=1 to get the orderly path, e.g. under a leak checker).
#include <cstdio>
#include <cstdlib>
#include <unistd.h>
extern "C" int yaha_main(int argc, char **argv);
int main(int argc, char **argv)
{
    const char *k = getenv("YAHA_KEEP_TEARDOWN"); const bool orderly = k && *k;
    if (!orderly) setenv("YAHA_FAST_EXIT", "1", 1);
    const int rc = yaha_main(argc, argv);
    fflush(stdout); fflush(stderr);
    if (!orderly) _exit(rc);
    return rc;
}
