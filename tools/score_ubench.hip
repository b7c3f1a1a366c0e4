// tools/score_ubench.hip — the gather + WARP score launch alone, on synthetic rows (a development harness, not part of the library):
// the kernels are the library's own (sbr_kernels.hip is included whole), launched through launch_score with SBR_SCORE_FORM
// selecting the form; times NREP launches with HIP events, reports the mean launch, the candidates scored per row and a checksum
// of NEGATIVES / TRIES (the forms must agree bit for bit).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-fast-math tools/score_ubench.hip -o tools/bin/score_ubench
//   tools/bin/score_ubench [rows] [items] [dim] [sigma]
#include "../sbr_rs_amd/csrc/sbr_kernels.hip"

#include <cmath>
#include <cstdio>
#include <random>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); return 1; } } while (0)

int main(int argc, char** argv) {
    const int R = argc > 1 ? atoi(argv[1]) : 255997;
    const uint32_t I = argc > 2 ? (uint32_t)atoll(argv[2]) : 1000000u;
    const int D = argc > 3 ? atoi(argv[3]) : 128;
    const float sigma = argc > 4 ? (float)atof(argv[4]) : 4.0f;  // std of a score: P(first candidate violates) = Phi(1 / (sigma sqrt 2))
    const int NSETS = 4, NREP = 40;
    std::mt19937_64 rng(12345);
    std::normal_distribution<float> nd(0.f, 1.f);
    std::vector<float> hE((size_t)I * D), hH((size_t)R * D), hb(I);
    const float es = sigma / std::sqrt((float)D);
    for (auto& x : hE) x = es * nd(rng);
    for (auto& x : hH) x = nd(rng);
    for (auto& x : hb) x = 0.01f * nd(rng);
    sbr::ModelView m{};
    m.d = D; m.ng = 4; m.num_items = I; m.loss = SBR_LOSS_WARP;
    CK(hipMalloc(&m.E, hE.size() * 4)); CK(hipMalloc(&m.b, hb.size() * 4));
    CK(hipMemcpy(m.E, hE.data(), hE.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(m.b, hb.data(), hb.size() * 4, hipMemcpyHostToDevice));
    sbr::BlockView blk{};
    CK(hipMalloc(&blk.H, hH.size() * 4));
    CK(hipMemcpy(blk.H, hH.data(), hH.size() * 4, hipMemcpyHostToDevice));
    CK(hipMalloc(&blk.in_idx, (size_t)R * 4)); CK(hipMalloc(&blk.out_idx, (size_t)R * 4)); CK(hipMalloc(&blk.neg, (size_t)R * 4)); CK(hipMalloc(&blk.coef, (size_t)R * 4));
    sbr::WorkView w{};
    CK(hipMalloc(&w.loss, (size_t)R * 4)); CK(hipMalloc(&w.tries, (size_t)R * 4)); CK(hipMalloc(&w.part_loss, 8192 * 8)); CK(hipMalloc(&w.part_tries, 8192 * 4));
    sbr::MbView mb[NSETS];
    std::vector<uint32_t> tmp(R);
    for (int s = 0; s < NSETS; ++s) {
        mb[s] = sbr::MbView{};
        mb[s].R = R; mb[s].B = 8192; mb[s].Tm = 64;
        uint32_t *a, *b2, *c;
        CK(hipMalloc(&a, (size_t)R * 4)); CK(hipMalloc(&b2, (size_t)R * 4)); CK(hipMalloc(&c, (size_t)R * 4));
        for (auto& x : tmp) x = (uint32_t)(rng() % I);
        CK(hipMemcpy(a, tmp.data(), (size_t)R * 4, hipMemcpyHostToDevice));
        for (auto& x : tmp) x = (uint32_t)(rng() % I);
        CK(hipMemcpy(b2, tmp.data(), (size_t)R * 4, hipMemcpyHostToDevice));
        for (int r = 0; r < R; ++r) tmp[r] = (uint32_t)(s * 16777216 + r);
        CK(hipMemcpy(c, tmp.data(), (size_t)R * 4, hipMemcpyHostToDevice));
        mb[s].in_idx = a; mb[s].out_idx = b2; mb[s].ctr = c;
    }
    hipStream_t st;
    CK(hipStreamCreate(&st));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 8; ++i) sbr::launch_score(m, mb[i % NSETS], blk, w, 0x1234567ull + i, R, st, nullptr);
    CK(hipStreamSynchronize(st));
    float total = 0.f, best = 1e9f;
    for (int i = 0; i < NREP; ++i) {
        CK(hipEventRecord(e0, st));
        sbr::launch_score(m, mb[i % NSETS], blk, w, 0x9999ull + 77 * i, R, st, nullptr);
        CK(hipEventRecord(e1, st));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        total += ms; best = ms < best ? ms : best;
    }
    // checksum of a fixed launch
    sbr::launch_score(m, mb[0], blk, w, 0xABCDEFull, R, st, nullptr);
    CK(hipStreamSynchronize(st));
    std::vector<uint32_t> neg(R), tr(R), oi(R), ii(R);
    std::vector<float> cf(R), ls(R);
    CK(hipMemcpy(neg.data(), blk.neg, (size_t)R * 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(tr.data(), w.tries, (size_t)R * 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(oi.data(), blk.out_idx, (size_t)R * 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(ii.data(), blk.in_idx, (size_t)R * 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(cf.data(), blk.coef, (size_t)R * 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(ls.data(), w.loss, (size_t)R * 4, hipMemcpyDeviceToHost));
    uint64_t ck = 1469598103934665603ull, ntries = 0;
    auto mix = [&](uint32_t v) { ck = (ck ^ v) * 1099511628211ull; };
    for (int r = 0; r < R; ++r) {
        mix(neg[r]); mix(tr[r]); mix(oi[r]); mix(ii[r]);
        uint32_t u; memcpy(&u, &cf[r], 4); mix(u); memcpy(&u, &ls[r], 4); mix(u);
        ntries += tr[r];
    }
    const double k = (double)ntries / R;
    const double bytes = (double)R * ((2.0 + k) * 4 * D + (1.0 + k) * 4);
#ifdef SBR_SCORE_PROF
    {
        unsigned long long z[8] = {0, 0, 0, 0, 0, 0, ~0ull, 0}, pr[8];
        CK(hipMemcpyToSymbol(HIP_SYMBOL(sbr::g_score_prof), z, sizeof z));
        sbr::launch_score(m, mb[1], blk, w, 0xABCDEull, R, st, nullptr);
        CK(hipStreamSynchronize(st));
        CK(hipMemcpyFromSymbol(pr, HIP_SYMBOL(sbr::g_score_prof), sizeof pr));
        const double wv = (double)pr[0];
        printf("prof: waves %.0f, rounds per wave %.2f, cycles per round: index work + requests %.0f, wait + tests + stores %.0f; wave lifetime mean %.0f max %llu cycles; first start to last end %.2f us\n",
               wv, pr[1] / wv, (double)pr[2] / pr[1], (double)pr[3] / pr[1], pr[4] / wv, pr[5], (pr[7] - pr[6]) / 100.0);
    }
#endif
    const char* form = getenv("SBR_SCORE_FORM");
    printf("form %s rows %d items %u d %d: mean %.2f us best %.2f us, k %.3f, algorithmic %.1f MB -> %.0f GB/s = %.3f of 8 TB/s (best %.3f); checksum %016llx\n",
           form ? form : "default", R, I, D, 1e3 * total / NREP, 1e3 * best, k, bytes / 1e6, bytes / (total / NREP * 1e-3) / 1e9,
           bytes / (total / NREP * 1e-3) / 8e12, bytes / (best * 1e-3) / 8e12, (unsigned long long)ck);
    return 0;
}
