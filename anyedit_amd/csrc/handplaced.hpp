// Helpers of the HAND-PLACED kernels (one wave per SIMD, csrc/ff_fused.hip, csrc/xattn_fused.hip): every MFMA is an asm statement — program order is kept and
// the register file of each accumulator is the constraint's ("v": VGPR, "a": the accumulator half) — followed by its slice of VALU work and a scheduling barrier.
// hipcc pads nothing around asm MFMAs and keeps no books for them: operands written by VALU must be a phase old when an MFMA reads them, results are read by
// VALU a phase later or after explicit nops, and operand fragments stay live (`hp_keep`) until two further MFMAs have been issued (the MFMA-source rule of
// tools/isa_audit.py::mfma_source_overwrites, asserted on the listings by tests/test_isa_static.py).
#pragma once
#include "common.hpp"
#include <type_traits>

__device__ __forceinline__ void hp_mfma_v(f32x4& acc, const u32x4& a, const u32x4& b) {
    asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b));
}
__device__ __forceinline__ void hp_mfma_v0(f32x4& acc, const u32x4& a, const u32x4& b) {       // C = 0
    asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, 0" : "=&v"(acc) : "v"(a), "v"(b));
}
__device__ __forceinline__ void hp_mfma_vc(f32x4& acc, const u32x4& a, const u32x4& b, const f32x4& c) {   // C = a register operand (e.g. a key mask)
    asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %3" : "=&v"(acc) : "v"(a), "v"(b), "v"(c));
}
__device__ __forceinline__ void hp_mfma_a(f32x4& acc, const u32x4& a, const u32x4& b) {
    asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc) : "v"(a), "v"(b));
}
__device__ __forceinline__ void hp_keep(const u32x4& v) { asm volatile("" ::"v"(v)); }

template <int I, int N, class F>
__device__ __forceinline__ void hp_static_for(F&& fn) {
    if constexpr (I < N) {
        fn(std::integral_constant<int, I>{});
        hp_static_for<I + 1, N>(fn);
    }
}

template <int N>
__device__ __forceinline__ void hp_wait_dma() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
