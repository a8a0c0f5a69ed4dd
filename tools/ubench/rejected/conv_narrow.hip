// 3x3 convolution with a NARROW output (Cout <= 4) for gfx950 (MI355X): the UNet's head, openaimodel.py:726-730 (`out[2]`: conv_nd(dims, model_channels,
// out_channels = 4, 3, padding=1)) and the VAE decoder's conv_out (diffusionmodules/model.py: 128 -> 3).
//
// Why a second kernel (round 5): the implicit-GEMM kernel (gemm_conv.hip) runs this layer on its smallest tile, 64 output columns wide — 4 of them real:
// 37 us per UNet evaluation at 29 TFLOP/s, fifteen sixteenths of its MFMA work multiplying zeros.  The layer is 9 * Cin * 4 multiply-adds per pixel
// (0.57 GFLOP at batch 12) on 31 MB of input: a byte-moving kernel.  Here: 64 pixels per block, four lanes per pixel (each a quarter of the 9 * Cin / 8
// sixteen-byte chunks of the pixel's 3x3 window: the four lanes of a pixel read 64 contiguous bytes per step), the whole weight tensor ([Cout][9][Cin] bf16,
// 23 KB for 320 -> 4) resident in LDS, v_dot2_f32_bf16 on the packed pairs as they arrive (fp32 accumulate, exact products), two cross-lane adds at the end.
// Halo taps are predicated loads (zero), as conv2d's zero padding.
#include "common.hpp"

namespace {

typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2v;

__device__ __forceinline__ float dot2(uint32_t a, uint32_t b, float c) {
#if defined(__HIP_DEVICE_COMPILE__)
    union { uint32_t u; bf16x2v v; } x, y;
    x.u = a; y.u = b;
    return __builtin_amdgcn_fdot2_f32_bf16(x.v, y.v, c, false);
#else
    return c;
#endif
}

struct NarrowArgs {
    const bf16_t* x; const bf16_t* w; const float* bias; void* y;
    int B, H, W, Cin, Cout, out_f32;
};

template <int NO>
__global__ __launch_bounds__(256) void conv3x3_narrow_kernel(const NarrowArgs p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];   // [NO][9][Cin] bf16
    u32x4* sw = reinterpret_cast<u32x4*>(smem_raw);
    const int nch = 9 * p.Cin / 8;                   // 16-byte chunks per output channel (Cin % 8 == 0)
    for (int i = threadIdx.x; i < NO * nch; i += 256) {
        const int co = i / nch;
        sw[i] = co < p.Cout ? reinterpret_cast<const u32x4*>(p.w)[i] : (u32x4){0u, 0u, 0u, 0u};
    }
    __syncthreads();
    const int M = p.B * p.H * p.W;
    const int m = blockIdx.x * 64 + (threadIdx.x >> 2), kq = threadIdx.x & 3;
    const int mc = min(m, M - 1);
    const int hw = p.H * p.W;
    const int b = mc / hw, rem = mc - b * hw;
    const int oy = rem / p.W, ox = rem - oy * p.W;
    const int cpt = p.Cin / 8;                       // chunks per tap
    float acc[NO];
#pragma unroll
    for (int o = 0; o < NO; ++o) acc[o] = 0.f;
    const bf16_t* xb = p.x + (long)b * hw * p.Cin;
    // the lane's chunks c = kq, kq + 4, ...: tap = c / cpt, channel offset (c % cpt) * 8; four loads in flight per lane
    for (int c0 = kq; c0 < nch; c0 += 16) {
        u32x4 xv[4];
        int cc[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int c = c0 + 4 * u;
            cc[u] = c;
            xv[u] = (u32x4){0u, 0u, 0u, 0u};
            if (c < nch) {
                const int tap = c / cpt, ci = (c - tap * cpt) * 8;
                const int iy = oy + tap / 3 - 1, ix = ox + tap % 3 - 1;
                if ((unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W)
                    xv[u] = *reinterpret_cast<const u32x4*>(xb + ((long)iy * p.W + ix) * p.Cin + ci);
            }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (cc[u] < nch) {
#pragma unroll
                for (int o = 0; o < NO; ++o) {
                    const u32x4 wv = sw[o * nch + cc[u]];
                    acc[o] = dot2(xv[u].x, wv.x, acc[o]);
                    acc[o] = dot2(xv[u].y, wv.y, acc[o]);
                    acc[o] = dot2(xv[u].z, wv.z, acc[o]);
                    acc[o] = dot2(xv[u].w, wv.w, acc[o]);
                }
            }
        }
    }
#pragma unroll
    for (int o = 0; o < NO; ++o) {   // the four lanes of a pixel: fixed order (q0 + q1) + (q2 + q3)
        acc[o] += __shfl_xor(acc[o], 1, 64);
        acc[o] += __shfl_xor(acc[o], 2, 64);
    }
    if (kq == 0 && m < M) {
#pragma unroll
        for (int o = 0; o < NO; ++o) {
            if (o < p.Cout) {
                const float v = acc[o] + (p.bias ? p.bias[o] : 0.f);
                if (p.out_f32) reinterpret_cast<float*>(p.y)[(long)m * p.Cout + o] = v;
                else reinterpret_cast<bf16_t*>(p.y)[(long)m * p.Cout + o] = f32_to_bf16(v);
            }
        }
    }
}

}  // namespace

// x [B,H,W,Cin] bf16 channels-last; w [Cout][9][Cin] bf16 ((ky, kx, cin) order: `ops.pack_conv3x3` with cin_pad = Cin); y [B,H,W,Cout] bf16 / fp32
extern "C" int ae_conv3x3_narrow_bf16(const void* x, const void* w, const float* bias, void* y, int B, int H, int W, int Cin, int Cout, int out_f32, void* stream) {
    AE_REQUIRE(x && w && y, "ae_conv3x3_narrow_bf16: null pointer");
    AE_REQUIRE(B > 0 && H > 0 && W > 0 && Cin > 0 && Cin % 8 == 0, "ae_conv3x3_narrow_bf16: bad shape B=%d H=%d W=%d Cin=%d (Cin %% 8 == 0)", B, H, W, Cin);
    AE_REQUIRE(Cout >= 1 && Cout <= 4, "ae_conv3x3_narrow_bf16: Cout=%d: this kernel is for 1..4 output channels (wider layers: ae_conv3x3_bf16)", Cout);
    AE_REQUIRE(((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(w)) & 15) == 0, "ae_conv3x3_narrow_bf16: x / w must be 16-byte aligned");
    const size_t lds = (size_t)4 * 9 * Cin * sizeof(bf16_t);
    AE_REQUIRE(lds <= 64 * 1024, "ae_conv3x3_narrow_bf16: Cin=%d: the weights must fit 64 KiB of LDS", Cin);
    AE_REQUIRE((long)B * H * W * Cin * 2 < (1L << 40), "ae_conv3x3_narrow_bf16: shape too large");
    NarrowArgs a{(const bf16_t*)x, (const bf16_t*)w, bias, y, B, H, W, Cin, Cout, out_f32};
    const long M = (long)B * H * W;
    hipLaunchKernelGGL(conv3x3_narrow_kernel<4>, dim3((unsigned)((M + 63) / 64)), dim3(256), lds, (hipStream_t)stream, a);
    return ae_check_launch("ae_conv3x3_narrow_bf16");
}
