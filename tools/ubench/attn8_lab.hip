// fp8 attention lab: standalone harness around anyedit_amd/csrc/attention_fp8.hip: fp64 CPU reference on the bf16 inputs (with and
// without the SAM rel-pos bias, ragged sizes), then timing at the SAM ViT-H global-attention shape next to the bf16 kernel.
#include "../../anyedit_amd/csrc/attention_fp8.hip"
#include "../../anyedit_amd/csrc/attention_fast.hip"
#include <cstdio>
#include <cmath>
#include <vector>
#include <random>
#include <cstring>
#include <stdarg.h>
void ae_set_error(const char* fmt, ...) { va_list ap; va_start(ap, fmt); vfprintf(stderr, fmt, ap); va_end(ap); fputc('\n', stderr); }
int ae_check_launch(const char* what) { hipError_t e = hipGetLastError(); if (e != hipSuccess) { fprintf(stderr, "%s: %s\n", what, hipGetErrorString(e)); return AE_ERR_LAUNCH; } return AE_OK; }
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
static uint16_t f2bf(float f) { uint32_t u; memcpy(&u, &f, 4); u += 0x7fffu + ((u >> 16) & 1u); return (uint16_t)(u >> 16); }
static float bf2f(uint16_t h) { uint32_t u = (uint32_t)h << 16; float f; memcpy(&f, &u, 4); return f; }

static void run(int B, int H, int Nq, int Nk, int D, bool bias, bool check, int iters, float qk_std) {
    const int C = H * D, N = std::max(Nq, Nk);
    std::mt19937 rng(99 + Nq + Nk);
    std::normal_distribution<float> nd(0.f, 1.f);
    std::vector<uint16_t> qkv((size_t)B * N * 3 * C);
    for (size_t i = 0; i < qkv.size(); ++i) qkv[i] = f2bf(nd(rng) * ((i / C) % 3 == 2 ? 1.0f : qk_std));
    const int kW = 64, kH = Nk / 64;
    std::vector<float> rh, rw;
    if (bias) { rh.resize((size_t)B * H * Nq * kH); rw.resize((size_t)B * H * Nq * kW); for (auto& x : rh) x = nd(rng); for (auto& x : rw) x = nd(rng); }
    uint16_t *dqkv, *dout; float *drh = nullptr, *drw = nullptr; char* ws;
    CK(hipMalloc(&dqkv, qkv.size() * 2)); CK(hipMalloc(&dout, (size_t)B * Nq * C * 2));
    CK(hipMemcpy(dqkv, qkv.data(), qkv.size() * 2, hipMemcpyHostToDevice));
    if (bias) { CK(hipMalloc(&drh, rh.size() * 4)); CK(hipMalloc(&drw, rw.size() * 4)); CK(hipMemcpy(drh, rh.data(), rh.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(drw, rw.data(), rw.size() * 4, hipMemcpyHostToDevice)); }
    const long wsb = ae_attn_fp8_workspace_bytes(B, H, Nq, Nk, D);
    CK(hipMalloc(&ws, wsb));
    CK(hipMemset(dout, 0xff, (size_t)B * Nq * C * 2));
    const float scale = 1.0f / std::sqrt((float)D);
    auto launch = [&]() {
        return ae_attn_fwd_fp8(dqkv, dqkv + C, dqkv + 2 * C, dout, B, H, Nq, Nk, D, (long)N * 3 * C, D, 3 * C, (long)N * 3 * C, D, 3 * C, (long)N * 3 * C, D, 3 * C,
                               (long)Nq * C, D, C, scale, drh, drw, kH, kW, ws, wsb, 0);
    };
    if (launch() != AE_OK) { printf("launch failed\n"); exit(1); }
    CK(hipDeviceSynchronize());
    if (check) {
        std::vector<uint16_t> out((size_t)B * Nq * C);
        CK(hipMemcpy(out.data(), dout, out.size() * 2, hipMemcpyDeviceToHost));
        double num = 0, den = 0, maxabs = 0;
        std::vector<double> sc(Nk);
        for (int b = 0; b < B; ++b) for (int h = 0; h < H; ++h) for (int i = 0; i < Nq; ++i) {
            const uint16_t* qr = &qkv[((size_t)b * N + i) * 3 * C + h * D];
            double mx = -1e300;
            for (int j = 0; j < Nk; ++j) {
                const uint16_t* kr = &qkv[((size_t)b * N + j) * 3 * C + C + h * D];
                double s = 0; for (int d = 0; d < D; ++d) s += (double)bf2f(qr[d]) * bf2f(kr[d]);
                s *= scale;
                if (bias) s += rh[(((size_t)b * H + h) * Nq + i) * kH + j / kW] + rw[(((size_t)b * H + h) * Nq + i) * kW + j % kW];
                sc[j] = s; mx = std::max(mx, s);
            }
            double l = 0; for (int j = 0; j < Nk; ++j) { sc[j] = std::exp(sc[j] - mx); l += sc[j]; }
            for (int d = 0; d < D; ++d) {
                double acc = 0;
                for (int j = 0; j < Nk; ++j) acc += sc[j] * bf2f(qkv[((size_t)b * N + j) * 3 * C + 2 * C + h * D + d]);
                acc /= l;
                const double got = bf2f(out[((size_t)b * Nq + i) * C + h * D + d]);
                num += (got - acc) * (got - acc); den += acc * acc; maxabs = std::max(maxabs, std::fabs(got - acc));
            }
        }
        const double rel = std::sqrt(num / den);
        printf("check fp8 B=%d H=%d Nq=%d Nk=%d D=%d bias=%d qk_std=%.1f : rel-L2 %.3e max-abs %.3e %s\n", B, H, Nq, Nk, D, bias, qk_std, rel, maxabs, (rel < 8e-2 && rel == rel) ? "OK" : "FAIL");
    }
    if (iters > 0) {
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        for (int i = 0; i < 3; ++i) launch();
        hipEventRecord(e0);
        for (int i = 0; i < iters; ++i) launch();
        hipEventRecord(e1);
        CK(hipDeviceSynchronize());
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double us = ms * 1e3 / iters, fl = 4.0 * B * H * (double)Nq * Nk * D;
        printf("time  fp8 (prepare + core) B=%d H=%d Nq=%d Nk=%d D=%d bias=%d : %8.1f us  %7.1f TFLOP/s\n", B, H, Nq, Nk, D, bias, us, fl / us / 1e6);
        if (!bias) {
            AttnArgs a{};
            a.q = dqkv; a.k = dqkv + C; a.v = dqkv + 2 * C; a.o = dout; a.B = B; a.H = H; a.Nq = Nq; a.Nk = Nk;
            a.q_sb = a.k_sb = a.v_sb = (long)N * 3 * C; a.q_sh = a.k_sh = a.v_sh = D; a.q_sn = a.k_sn = a.v_sn = 3 * C;
            a.o_sb = (long)Nq * C; a.o_sh = D; a.o_sn = C; a.scale = scale;
            for (int i = 0; i < 3; ++i) ae_attn_fast_launch(a, D, 0);
            hipEventRecord(e0);
            for (int i = 0; i < iters; ++i) ae_attn_fast_launch(a, D, 0);
            hipEventRecord(e1);
            CK(hipDeviceSynchronize());
            hipEventElapsedTime(&ms, e0, e1);
            printf("time  bf16 fast kernel, same shape                                   : %8.1f us  %7.1f TFLOP/s\n", ms * 1e3 / iters, fl / (ms * 1e3 / iters) / 1e6);
        }
    }
    hipFree(dqkv); hipFree(dout); hipFree(ws); if (drh) hipFree(drh); if (drw) hipFree(drw);
}

int main() {
    run(1, 2, 256, 256, 80, false, true, 0, 1.0f);
    run(1, 2, 200, 190, 80, false, true, 0, 1.0f);
    run(2, 2, 130, 320, 80, false, true, 0, 2.0f);
    run(1, 2, 384, 4096, 80, true, true, 0, 1.0f);
    run(1, 1, 192, 4096, 80, true, true, 0, 2.0f);
    run(1, 16, 4096, 4096, 80, false, false, 20, 1.0f);
    run(1, 16, 4096, 4096, 80, true, false, 20, 1.0f);
    run(12, 8, 1024, 1024, 80, false, false, 20, 1.0f);
    return 0;
}
