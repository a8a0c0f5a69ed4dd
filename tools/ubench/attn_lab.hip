// Attention lab: standalone harness around anyedit_amd/csrc/attention_fast.hip for the optimisation loop (no Python, no torch:
// a gpurun visit costs seconds).  1) prints the lane maps of ds_read_b64_tr_b16 and v_permlane16_swap the kernel relies on,
// 2) checks the kernel against an fp64 CPU reference at small sizes (tail / ragged / spiked cases), 3) times the UNet shapes.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffinite-math-only -I anyedit_amd/csrc -o tools/ubench/build/attn_lab tools/ubench/attn_lab.hip
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

__global__ void probe(unsigned* o, const unsigned* in) {
    __shared__ __attribute__((aligned(16))) unsigned short sm[1024];
    for (int i = threadIdx.x; i < 1024; i += 64) sm[i] = (unsigned short)i;
    __syncthreads();
    const s16x4v r = lds_tr16((const char*)(sm + threadIdx.x * 4));
    for (int j = 0; j < 4; ++j) o[threadIdx.x * 4 + j] = (unsigned short)r[j];
    const u32x2 s = __builtin_amdgcn_permlane16_swap(in[threadIdx.x], in[threadIdx.x + 64], false, false);
    o[256 + threadIdx.x] = s[0];
    o[320 + threadIdx.x] = s[1];
}

struct Case { int B, H, Nq, Nk, D; int spike; int Nk2 = 0; };

// fused-qkv layout as CrossAttention uses it: rows [B*N, 3*H*D], q | k | v column blocks
static double run_case(const Case& cs, bool check, int iters, double* us_out) {
    const int B = cs.B, H = cs.H, Nq = cs.Nq, Nk = cs.Nk, D = cs.D, C = H * D;
    const int N = Nq > Nk ? Nq : Nk;
    std::vector<uint16_t> qkv((size_t)B * N * 3 * C);
    std::mt19937 rng(1234 + Nq * 7 + Nk);
    std::normal_distribution<float> nd(0.f, 1.f);
    for (auto& x : qkv) x = f2bf(nd(rng) * 1.5f);
    if (cs.spike) {  // one key row strongly aligned with one query row late in the sequence: forces the rebase branch mid-stream
        for (int bh = 0; bh < B * H; ++bh) {
            const int b = bh / H, h = bh % H;
            const int qr = (7 + 13 * bh) % Nq, kr = Nk - 1 - ((5 * bh) % (Nk / 2));
            for (int d = 0; d < D; ++d) {
                const float qv = bf2f(qkv[((size_t)b * N + qr) * 3 * C + h * D + d]);
                qkv[((size_t)b * N + kr) * 3 * C + C + h * D + d] = f2bf(qv * 6.0f);
            }
        }
    }
    const int Nk2 = cs.Nk2;
    std::vector<uint16_t> kv2((size_t)B * std::max(Nk2, 1) * 2 * C);
    for (auto& x : kv2) x = f2bf(nd(rng) * 1.5f);
    std::vector<float> sc2(B);
    for (int i = 0; i < B; ++i) sc2[i] = 0.25f + 0.5f * i;
    uint16_t* dkv2; float* dsc2;
    CK(hipMalloc(&dkv2, kv2.size() * 2)); CK(hipMalloc(&dsc2, B * 4));
    CK(hipMemcpy(dkv2, kv2.data(), kv2.size() * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(dsc2, sc2.data(), B * 4, hipMemcpyHostToDevice));
    uint16_t *dqkv, *dout;
    CK(hipMalloc(&dqkv, qkv.size() * 2));
    CK(hipMalloc(&dout, (size_t)B * Nq * C * 2));
    CK(hipMemcpy(dqkv, qkv.data(), qkv.size() * 2, hipMemcpyHostToDevice));
    CK(hipMemset(dout, 0xff, (size_t)B * Nq * C * 2));
    AttnArgs a{};
    a.q = dqkv; a.k = dqkv + C; a.v = dqkv + 2 * C; a.o = dout;
    a.B = B; a.H = H; a.Nq = Nq; a.Nk = Nk;
    a.q_sb = a.k_sb = a.v_sb = (long)N * 3 * C; a.q_sh = a.k_sh = a.v_sh = D; a.q_sn = a.k_sn = a.v_sn = 3 * C;
    a.o_sb = (long)Nq * C; a.o_sh = D; a.o_sn = C;
    a.scale = 1.0f / std::sqrt((float)D);
    if (Nk2) { a.k2 = dkv2; a.v2 = dkv2 + C; a.Nk2 = Nk2; a.k2_sb = a.v2_sb = (long)Nk2 * 2 * C; a.k2_sh = a.v2_sh = D; a.k2_sn = a.v2_sn = 2 * C; a.scale2 = dsc2; }
    if (getenv("AE_LAB_NOZERO")) a.kH = 12345;
    if (getenv("AE_LAB_CLOCK")) a.kW = 777;
    int rc = ae_attn_fast_launch(a, D, 0);
    if (rc != AE_OK) { printf("launch rc=%d\n", rc); exit(1); }
    CK(hipDeviceSynchronize());
    double rel = 0.0;
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
                sc[j] = s * a.scale; mx = std::max(mx, sc[j]);
            }
            double l = 0; for (int j = 0; j < Nk; ++j) { sc[j] = std::exp(sc[j] - mx); l += sc[j]; }
            std::vector<double> s2(std::max(Nk2, 1)); double l2 = 0;
            if (Nk2) {
                double m2 = -1e300;
                for (int j = 0; j < Nk2; ++j) {
                    const uint16_t* kr = &kv2[((size_t)b * Nk2 + j) * 2 * C + h * D];
                    double t = 0; for (int d = 0; d < D; ++d) t += (double)bf2f(qr[d]) * bf2f(kr[d]);
                    s2[j] = t * a.scale; m2 = std::max(m2, s2[j]);
                }
                for (int j = 0; j < Nk2; ++j) { s2[j] = std::exp(s2[j] - m2); l2 += s2[j]; }
            }
            for (int d = 0; d < D; ++d) {
                double acc = 0;
                for (int j = 0; j < Nk; ++j) acc += sc[j] * bf2f(qkv[((size_t)b * N + j) * 3 * C + 2 * C + h * D + d]);
                acc /= l;
                if (Nk2) {
                    double a2 = 0;
                    for (int j = 0; j < Nk2; ++j) a2 += s2[j] * bf2f(kv2[((size_t)b * Nk2 + j) * 2 * C + C + h * D + d]);
                    acc += sc2[b] * a2 / l2;
                }
                const double got = bf2f(out[((size_t)b * Nq + i) * C + h * D + d]);
                num += (got - acc) * (got - acc); den += acc * acc; maxabs = std::max(maxabs, std::fabs(got - acc));
            }
        }
        rel = std::sqrt(num / den);
        printf("check B=%d H=%d Nq=%d Nk=%d D=%d spike=%d Nk2=%d : rel-L2 %.3e  max-abs %.3e  %s\n", B, H, Nq, Nk, D, cs.spike, Nk2, rel, maxabs, (rel < 6e-3 && rel == rel) ? "OK" : "FAIL");
    }
    if (iters > 0) {
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        for (int i = 0; i < 3; ++i) ae_attn_fast_launch(a, D, 0);
        { unsigned long long z[4] = {0, 0, 0, 0}; CK(hipMemcpyToSymbol(HIP_SYMBOL(g_attn_dbg), z, sizeof(z))); }
        hipEventRecord(e0);
        for (int i = 0; i < iters; ++i) ae_attn_fast_launch(a, D, 0);
        hipEventRecord(e1);
        CK(hipDeviceSynchronize());
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double us = ms * 1e3 / iters, fl = 4.0 * B * H * (double)Nq * Nk * D;
        unsigned long long dbg[4];
        CK(hipMemcpyFromSymbol(dbg, HIP_SYMBOL(g_attn_dbg), sizeof(dbg)));
        const double cyc_per_blk = (double)dbg[0] / dbg[2], ns_per_blk = (double)dbg[1] / dbg[2] * 10.0;
        printf("time  B=%d H=%d Nq=%d Nk=%d D=%d Nk2=%d : %8.1f us  %7.1f TFLOP/s (un-padded d)  = %.3f of 2.5 PF | per block %.0f cycles, %.0f ns -> %.2f GHz\n", B, H, Nq, Nk, D, Nk2, us, fl / us / 1e6, fl / us / 1e6 / 2500.0, cyc_per_blk, ns_per_blk, cyc_per_blk / ns_per_blk);
        if (us_out) *us_out = us;
    }
    hipFree(dqkv); hipFree(dout); hipFree(dkv2); hipFree(dsc2);
    return rel;
}

int main(int argc, char** argv) {
    const bool do_probe = argc > 1 && strchr(argv[1], 'p');
    const bool do_check = argc <= 1 || strchr(argv[1], 'c');
    const bool do_time = argc <= 1 || strchr(argv[1], 't');
    if (do_probe) {
        unsigned *d_o, *d_in; unsigned h_in[128], h_o[384];
        for (int i = 0; i < 128; ++i) h_in[i] = i;
        CK(hipMalloc(&d_o, sizeof(h_o))); CK(hipMalloc(&d_in, sizeof(h_in)));
        CK(hipMemcpy(d_in, h_in, sizeof(h_in), hipMemcpyHostToDevice));
        hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d_o, d_in);
        CK(hipDeviceSynchronize());
        CK(hipMemcpy(h_o, d_o, sizeof(h_o), hipMemcpyDeviceToHost));
        printf("ds_read_b64_tr_b16, lane L addresses elements 4L..4L+3; received (lane: e0 e1 e2 e3):\n");
        for (int l = 0; l < 64; ++l) printf("%2d:%4u%4u%4u%4u%s", l, h_o[l * 4], h_o[l * 4 + 1], h_o[l * 4 + 2], h_o[l * 4 + 3], (l % 4 == 3) ? "\n" : "   ");
        printf("v_permlane16_swap(vdst = lane id, src0 = 64 + lane id): new vdst / new src0\n");
        for (int l = 0; l < 64; ++l) printf("%3u%s", h_o[256 + l], (l % 16 == 15) ? "\n" : " ");
        for (int l = 0; l < 64; ++l) printf("%3u%s", h_o[320 + l], (l % 16 == 15) ? "\n" : " ");
    }
    if (do_check) {
        const Case cases[] = {{1, 2, 128, 128, 40, 0}, {1, 2, 256, 320, 40, 0}, {2, 3, 200, 200, 40, 0}, {1, 2, 384, 77, 40, 0}, {1, 2, 512, 512, 40, 1},
                              {1, 2, 256, 256, 80, 0}, {1, 1, 130, 190, 80, 1}, {2, 2, 300, 78, 40, 0, 4}, {2, 2, 256, 78, 80, 0, 4}, {1, 2, 200, 33, 40, 0, 40}, {1, 2, 64, 16, 40, 0}};
        for (const auto& cs : cases) run_case(cs, true, 0, nullptr);
    }
    if (do_time) {
        run_case({12, 8, 4096, 4096, 40, 0}, false, 20, nullptr);
        run_case({12, 8, 1024, 1024, 80, 0}, false, 20, nullptr);
        run_case({12, 8, 4096, 78, 40, 0}, false, 20, nullptr);
        run_case({12, 8, 4096, 78, 40, 0, 4}, false, 20, nullptr);
        run_case({12, 8, 1024, 78, 80, 0, 4}, false, 20, nullptr);
    }
    return 0;
}
