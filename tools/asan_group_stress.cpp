// Host-memory stress of the block and group calls of include/mmc_hip.h without Python: random small sequences,
// first as one block, then cut into random numbers of frame blocks through mmc_group_*; both results must be
// equal bit for bit.  Meant to be built together with the library sources under the host AddressSanitizer
// (tools/asan_build.sh), so that a heap error in the library names its place.
// usage: asan_group_stress [cases] [seed]
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <unistd.h>
#include <vector>
#include "../include/mmc_hip.h"

#define CK(call)                                                                                     \
    do {                                                                                             \
        int rc_ = (call);                                                                            \
        if (rc_ != MMC_OK) { fprintf(stderr, "case %d: %s failed (%d): %s | %s\n", n, #call, rc_,    \
                                     mmc_last_error(hb), mmc_group_last_error(gp)); return 2; }     \
    } while (0)

int main(int argc, char **argv) {
    const int cases = argc > 1 ? atoi(argv[1]) : 100;
    std::mt19937 rng(argc > 2 ? (unsigned)atoi(argv[2]) : 1u);
    auto ri = [&](int lo, int hi) { return lo + (int)(rng() % (unsigned)(hi - lo)); };          // [lo, hi)
    auto rf = [&]() { return (float)(rng() >> 8) / (float)(1u << 24); };                           // [0, 1)
    // STRESS_NAP_MS: idle time around mmc_group_create, as an interpreter between the calls would leave it
    const unsigned nap_us = getenv("STRESS_NAP_MS") ? 1000u * (unsigned)atoi(getenv("STRESS_NAP_MS")) : 0u;
    int bad = 0;
    for (int n = 0; n < cases; ++n) {
        const int factors[] = {4, 4, 2, 3};
        const int factor = factors[ri(0, 4)];
        const int ny = ri(6, 60), nx = ri(6, 80), nc = ri(2, 10), nblocks = ri(2, nc + 1);
        const int sampling = (factor == 4 && ri(0, 10) < 6) ? MMC_SAMPLING_AVERAGE : MMC_SAMPLING_STRIDE;
        const int ks[] = {0, 3, 5, 7};
        const int kk = ks[ri(0, 4)];
        const float kappas[] = {0.25f, 1.f, NAN, 0.6f};
        const float kappa = kappas[ri(0, 4)];
        const int direction = ri(0, 3);
        const float scale = (float[]){0.7f, 2.5f, 8.f}[ri(0, 3)];
        const bool with_occ = ri(0, 10) < 4, outliers = ri(0, 10) < 4, device_start = ri(0, 2) == 0;
        const int iters = ri(5, 30), check = ri(0, 2) ? 7 : 0;
        const int Ny = ny * factor, Nx = nx * factor;
        const size_t N = (size_t)Ny * Nx, nlr = (size_t)ny * nx, np = (size_t)nc - 1;
        std::vector<float> f(nlr * nc), v(N * np * 2), k((size_t)kk * kk), u0(N * nc);
        std::vector<uint8_t> occ(N * np, 0);
        for (float &x : f) x = rf();
        for (float &x : v) x = (rf() - 0.5f) * 2.f * scale;
        if (outliers) for (size_t i = 0; i < v.size(); ++i) if (rng() % 500 == 0) v[i] += 150.f;
        if (with_occ) for (uint8_t &x : occ) x = rng() % 10 == 0;
        float ksum = 0.f;
        if (kk > 0 && ri(0, 2) == 0) {
            // an outer product: the library applies it as two 1-D passes where it is large enough
            std::vector<float> ka((size_t)kk), kb((size_t)kk);
            for (float &x : ka) x = 0.2f + rf();
            for (float &x : kb) x = 0.2f + rf();
            for (int j = 0; j < kk; ++j) for (int i = 0; i < kk; ++i) { k[(size_t)i + (size_t)kk * j] = ka[i] * kb[j]; ksum += ka[i] * kb[j]; }
        } else {
            for (float &x : k) { x = 0.2f + rf(); ksum += x; }
        }
        for (float &x : k) x /= ksum;
        for (float &x : u0) x = rf();
        fprintf(stderr, "case %d: ny=%d nx=%d nc=%d blocks=%d f=%d sampling=%d k=%d kappa=%g dir=%d occ=%d outliers=%d dev_start=%d iters=%d\n",
                n, ny, nx, nc, nblocks, factor, sampling, kk, kappa, direction, with_occ, outliers, device_start, iters);

        mmc_params p;
        memset(&p, 0, sizeof(p));
        p.size = sizeof(p); p.version = MMC_ABI_VERSION;
        p.ny = ny; p.nx = nx; p.factor = factor; p.nc_global = nc; p.frame_begin = 0; p.frame_end = nc;
        p.alpha = 0.01f; p.kappa = kappa; p.h = 0.f; p.flow_direction = direction; p.sampling = sampling; p.device = 0;
        const bool has_w = !(std::isnan(kappa) || kappa == 1.f);
        std::vector<float> u1(N * nc), w1(N * nc), ug(N * nc), wg(N * nc);
        mmc_handle *hb = nullptr;
        mmc_group *gp = nullptr;
        float h1 = 0.f, hg = 0.f;
        mmc_stats st1, stg;
        CK(mmc_create(&p, &hb));
        CK(mmc_set_lowres(hb, f.data()));
        CK(mmc_set_blur(hb, kk ? k.data() : nullptr, kk, kk));
        CK(mmc_set_flow(hb, v.data(), with_occ ? occ.data() : nullptr));
        CK(device_start ? mmc_set_state(hb, nullptr, nullptr) : mmc_set_state(hb, u0.data(), has_w ? u0.data() : nullptr));
        CK(mmc_init(hb));
        CK(mmc_get_h(hb, &h1));
        CK(mmc_solve(hb, iters, 0.f, check, &st1, nullptr, nullptr));
        CK(mmc_get_state(hb, u1.data(), has_w ? w1.data() : nullptr));
        CK(mmc_destroy(hb));
        hb = nullptr;

        if (nap_us) usleep(nap_us);
        CK(mmc_group_create(&p, nblocks, nullptr, &gp));
        if (nap_us) usleep(nap_us);
        CK(mmc_group_set_lowres(gp, f.data()));
        CK(mmc_group_set_blur(gp, kk ? k.data() : nullptr, kk, kk));
        CK(mmc_group_set_flow(gp, v.data(), with_occ ? occ.data() : nullptr));
        CK(device_start ? mmc_group_set_state(gp, nullptr, nullptr) : mmc_group_set_state(gp, u0.data(), has_w ? u0.data() : nullptr));
        CK(mmc_group_init(gp));
        CK(mmc_group_get_h(gp, &hg));
        CK(mmc_group_set_h(gp, h1));
        CK(mmc_group_solve(gp, iters, 0.f, check, &stg, nullptr, nullptr));
        CK(mmc_group_get_state(gp, ug.data(), has_w ? wg.data() : nullptr));
        CK(mmc_group_destroy(gp));
        gp = nullptr;
        const bool same = memcmp(u1.data(), ug.data(), u1.size() * sizeof(float)) == 0 &&
                          (!has_w || memcmp(w1.data(), wg.data(), w1.size() * sizeof(float)) == 0) &&
                          std::fabs(h1 - hg) <= 1e-6f * std::fabs(h1);
        if (!same) { ++bad; fprintf(stderr, "MISS in case %d (h %.7g / %.7g)\n", n, h1, hg); }
    }
    printf("cases %d, misses %d\n", cases, bad);
    fflush(stdout);       // the sanitizer and the HIP runtime may disagree about the order of their exit handlers
    return bad ? 1 : 0;
}
