// scripts/exp/startup.cpp -- where a drop-in block's start-up time goes (C1 through a pipe is 97 % start-up:
// profiles/r03_c1pipe.json): timestamps around the first HIP call, filter creation, pinned buffers, first and
// second call.  g++ -O2 startup.cpp -I../../include -L../../sdr_amd/lib -lsdr_amd -o startup
#include <chrono>
#include <cstdio>
#include <vector>

#include "sdr_amd.h"

static double now() {
    using namespace std::chrono;
    return duration<double>(steady_clock::now().time_since_epoch()).count();
}

int main() {
    const double t0 = now();
    const int ndev = sdr_device_count();
    const double t1 = now();
    std::vector<float> h(127);
    sdr_hilbert_taps(h.data(), h.size());
    sdr_fir_t* f = nullptr;
    int rc = sdr_fir_create(&f, h.data(), h.size(), SDR_REAL, 1, 0);
    const double t2 = now();
    void* a = sdr_host_alloc(8u << 20);
    void* b = sdr_host_alloc(8u << 20);
    void* c = sdr_host_alloc(8u << 20);
    void* d = sdr_host_alloc(8u << 20);
    const double t3 = now();
    rc |= sdr_fir_process(f, a, 2044, b);
    const double t4 = now();
    rc |= sdr_fir_process(f, a, 2044, b);
    const double t5 = now();
    rc |= sdr_fir_process(f, c, 2u << 20, d);
    const double t6 = now();
    sdr_fir_destroy(f);
    sdr_host_free(a); sdr_host_free(b); sdr_host_free(c); sdr_host_free(d);
    const double t7 = now();
    printf("devices %d rc %d | first HIP call %.1f ms | create %.1f ms | 4 x 8 MiB pinned %.1f ms | first call %.1f ms | second call %.3f ms | "
           "2 Mi-sample call %.3f ms | teardown %.1f ms | total %.1f ms\n",
           ndev, rc, 1e3 * (t1 - t0), 1e3 * (t2 - t1), 1e3 * (t3 - t2), 1e3 * (t4 - t3), 1e3 * (t5 - t4), 1e3 * (t6 - t5), 1e3 * (t7 - t6),
           1e3 * (t7 - t0));
    return rc;
}
