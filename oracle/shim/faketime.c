/* oracle/shim/faketime.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 * LD_PRELOAD shim for the reference's prebuilt SNAP binary (gem/c_exe/node2vec, call site gem/embedding/node2vec.py:34-48):
 * the binary seeds both of its TRnd generators with time(NULL) (node2vec() and LearnEmbeddings(): `time@plt`, TRnd::PutSeed
 * @0x41b9a0 stores a non-zero seed as it is).  With time() pinned and OMP_NUM_THREADS=1 the binary is DETERMINISTIC, which is
 * what lets oracle/snap_stream.py be checked against it walk for walk (scripts/make_golden_n2v_snap_stream.py).
 * Built into oracle/_ref/libfaketime.so by oracle/Makefile; kept out of liboracle.so on purpose (it must never shadow time()
 * in a process that did not ask for it). */
#define _GNU_SOURCE
#include <stdlib.h>
#include <time.h>

time_t time(time_t *t)
{
    const char *e = getenv("GEM_FAKE_TIME");
    const time_t v = e ? (time_t)atoll(e) : (time_t)1000;
    if (t) *t = v;
    return v;
}

/* gem/c_src/gf.cpp:46 seeds its embedding generator with std::chrono::system_clock::now() (clock_gettime(CLOCK_REALTIME) inside libstdc++), not
 * with time(): GEM_FAKE_CLOCK="<sec>[.<nsec>]" freezes that clock as well (only when set: the node2vec goldens were made without it), which makes
 * oracle/_ref/gf deterministic -- scripts/make_golden_gf_cpp.py, tests/test_oracle_gf.py. */
#include <dlfcn.h>
int clock_gettime(clockid_t id, struct timespec *ts)
{
    const char *e = getenv("GEM_FAKE_CLOCK");
    if (e && id == CLOCK_REALTIME && ts) {
        char *end = NULL;
        ts->tv_sec = (time_t)strtoll(e, &end, 10);
        ts->tv_nsec = (end && *end == '.') ? atol(end + 1) : 0;
        return 0;
    }
    static int (*real)(clockid_t, struct timespec *) = NULL;
    if (!real) real = (int (*)(clockid_t, struct timespec *))dlsym(RTLD_NEXT, "clock_gettime");
    return real ? real(id, ts) : -1;
}
