// Probe of the gfx950 MX-scaled fp8 MFMA (v_mfma_scale_f32_32x32x64_f8f6f4 / 16x16x128): operand lane maps, scale semantics, fp8 conversion.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <cstring>
#include <vector>
#include <random>
typedef __attribute__((ext_vector_type(8))) int i32x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

__global__ void k32(float* o, const unsigned char* a, const unsigned char* b, int sa, int sb) {  // a: [32][64] row-major fp8, b: [32 cols][64] (B^T row-major)
    const int l = threadIdx.x;
    i32x8 A, B;
    const int* ap = (const int*)(a + (l & 31) * 64 + (l >> 5) * 32);
    const int* bp = (const int*)(b + (l & 31) * 64 + (l >> 5) * 32);
    for (int i = 0; i < 8; ++i) { A[i] = ap[i]; B[i] = bp[i]; }
    f32x16 c; for (int i = 0; i < 16; ++i) c[i] = 0.f;
    c = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(A, B, c, 0, 0, 0, sa, 0, sb);
    for (int r = 0; r < 16; ++r) o[((r & 3) + 8 * (r >> 2) + 4 * (l >> 5)) * 32 + (l & 31)] = c[r];   // row-major [32][32]
}
__global__ void k16(float* o, const unsigned char* a, const unsigned char* b, int sa, int sb) {  // a: [16][128], b: [16][128]
    const int l = threadIdx.x;
    i32x8 A, B;
    const int* ap = (const int*)(a + (l & 15) * 128 + (l >> 4) * 32);
    const int* bp = (const int*)(b + (l & 15) * 128 + (l >> 4) * 32);
    for (int i = 0; i < 8; ++i) { A[i] = ap[i]; B[i] = bp[i]; }
    f32x4 c = {0, 0, 0, 0};
    c = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(A, B, c, 0, 0, 0, sa, 0, sb);
    for (int r = 0; r < 4; ++r) o[(4 * (l >> 4) + r) * 16 + (l & 15)] = c[r];
}
__global__ void kcvt(unsigned* o, const float* x) {
    const int l = threadIdx.x;
    int pk = __builtin_amdgcn_cvt_pk_fp8_f32(x[4 * l], x[4 * l + 1], 0, false);
    pk = __builtin_amdgcn_cvt_pk_fp8_f32(x[4 * l + 2], x[4 * l + 3], pk, true);
    o[l] = (unsigned)pk;
}
static float f8tof(unsigned char v) {  // OCP e4m3fn
    const int s = v >> 7, e = (v >> 3) & 15, m = v & 7;
    float f = e == 0 ? std::ldexp((float)m / 8.f, -6) : std::ldexp(1.f + m / 8.f, e - 7);
    return s ? -f : f;
}
int main() {
    std::mt19937 rng(3);
    std::vector<unsigned char> a(32 * 64), b(32 * 64), a2(16 * 128), b2(16 * 128);
    auto rnd8 = [&]() { unsigned char v; do { v = (unsigned char)(rng() & 0xff); } while ((v & 0x7f) > 0x50); return v; };  // |x| <= 8
    for (auto& x : a) x = rnd8(); for (auto& x : b) x = rnd8(); for (auto& x : a2) x = rnd8(); for (auto& x : b2) x = rnd8();
    unsigned char *da, *db; float* dout;
    CK(hipMalloc(&da, 4096)); CK(hipMalloc(&db, 4096)); CK(hipMalloc(&dout, 4096 * 4));
    for (int sc = 0; sc < 2; ++sc) {
        const int sa = 127, sb = sc ? 129 : 127;  // E8M0: 2^(x-127)
        CK(hipMemcpy(da, a.data(), a.size(), hipMemcpyHostToDevice)); CK(hipMemcpy(db, b.data(), b.size(), hipMemcpyHostToDevice));
        hipLaunchKernelGGL(k32, dim3(1), dim3(64), 0, 0, dout, da, db, sa, sb);
        std::vector<float> o(1024); CK(hipMemcpy(o.data(), dout, 4096, hipMemcpyDeviceToHost));
        double err = 0, ref_n = 0;
        for (int i = 0; i < 32; ++i) for (int j = 0; j < 32; ++j) {
            double s = 0; for (int k = 0; k < 64; ++k) s += (double)f8tof(a[i * 64 + k]) * f8tof(b[j * 64 + k]);
            s *= sc ? 4.0 : 1.0;
            err += std::fabs(o[i * 32 + j] - s); ref_n += std::fabs(s);
        }
        printf("32x32x64 scale_b=%d: sum|err| %.4g of sum|ref| %.4g\n", sb, err, ref_n);
        CK(hipMemcpy(da, a2.data(), a2.size(), hipMemcpyHostToDevice)); CK(hipMemcpy(db, b2.data(), b2.size(), hipMemcpyHostToDevice));
        hipLaunchKernelGGL(k16, dim3(1), dim3(64), 0, 0, dout, da, db, sa, sb);
        CK(hipMemcpy(o.data(), dout, 1024, hipMemcpyDeviceToHost));
        err = 0; ref_n = 0;
        for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) {
            double s = 0; for (int k = 0; k < 128; ++k) s += (double)f8tof(a2[i * 128 + k]) * f8tof(b2[j * 128 + k]);
            s *= sc ? 4.0 : 1.0;
            err += std::fabs(o[i * 16 + j] - s); ref_n += std::fabs(s);
        }
        printf("16x16x128 scale_b=%d: sum|err| %.4g of sum|ref| %.4g\n", sb, err, ref_n);
    }
    float hx[256]; for (int i = 0; i < 256; ++i) hx[i] = (i - 100) * 0.37f;
    float* dx; unsigned* dp; CK(hipMalloc(&dx, 1024)); CK(hipMalloc(&dp, 256)); CK(hipMemcpy(dx, hx, 1024, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(kcvt, dim3(1), dim3(64), 0, 0, dp, dx);
    unsigned hp[64]; CK(hipMemcpy(hp, dp, 256, hipMemcpyDeviceToHost));
    printf("cvt_pk_fp8_f32: x -> fp8 -> float (lane 0..3):\n");
    for (int l = 0; l < 4; ++l) for (int j = 0; j < 4; ++j) printf("  %8.3f -> 0x%02x = %8.3f\n", hx[4 * l + j], (hp[l] >> (8 * j)) & 0xff, f8tof((hp[l] >> (8 * j)) & 0xff));
    for (int l = 40; l < 42; ++l) for (int j = 0; j < 4; ++j) printf("  %8.3f -> 0x%02x = %8.3f\n", hx[4 * l + j], (hp[l] >> (8 * j)) & 0xff, f8tof((hp[l] >> (8 * j)) & 0xff));
    return 0;
}
