// Device-side helpers shared by the HIP kernels of libtronhip (Kaiser-Bessel window, sample loads).
#pragma once

#include <hip/hip_fp16.h>
#include <hip/hip_runtime.h>

#include "../../include/tron_hip.h"
#include "tron_internal.h"

namespace tron {

// ------------------------------------------------------------------------- Kaiser-Bessel

// src/tron.cu:304-321, op for op: the coefficient literals are doubles, so both Horner chains
// run in double (unfused) and are rounded to float; the quotient is an IEEE float division.
__device__ __forceinline__ float besseli0_ref(const float x)
{
    if (x == 0.f) return 1.f;
    float z = x * x;
    float num = (z* (z* (z* (z* (z* (z* (z* (z* (z* (z* (z* (z* (z*
        (z* 0.210580722890567e-22  + 0.380715242345326e-19 ) +
        0.479440257548300e-16) + 0.435125971262668e-13 ) +
        0.300931127112960e-10) + 0.160224679395361e-7  ) +
        0.654858370096785e-5)  + 0.202591084143397e-2  ) +
        0.463076284721000e0)   + 0.754337328948189e2   ) +
        0.830792541809429e4)   + 0.571661130563785e6   ) +
        0.216415572361227e8)   + 0.356644482244025e9   ) +
        0.144048298227235e10);
    float den = (z*(z*(z-0.307646912682801e4)+
        0.347626332405882e7)-0.144048298227235e10);
    return -num/den;
}

struct KbCoef {
    float W, invW, beta;
    float poly[kKbPolyTerms];
};

// src/tron.cu:338-349.  EXACT: the reference's expression tree.  FAST: (0.5/W)*I0(beta*sqrt(s)) is
// an entire function of s = 1-(x/W)^2; a fixed-degree polynomial in s (Chebyshev-economised on
// the host, coefficients held in scalar registers) replaces sqrt, the rational I0 and the division.
template <int KB>
__device__ __forceinline__ float kb_weight(const float x, const KbCoef &k)
{
    if (!(fabsf(x) < k.W)) return 0.0f;
    if (KB == TRON_KB_EXACT) {
        float r = x / k.W;
        float f = sqrtf(1.0f - r * r);
        return 0.5f * besseli0_ref(k.beta * f) / k.W;
    } else {
        float r = x * k.invW;
        float s = fmaf(-r, r, 1.0f);
        float acc = k.poly[0];
#pragma unroll
        for (int t = 1; t < kKbPolyTerms; ++t) acc = fmaf(acc, s, k.poly[t]);
        return acc;
    }
}

typedef float v2f __attribute__((ext_vector_type(2)));

// Two Kaiser-Bessel weights at once (v_pk_fma_f32): component-wise the same fmaf chain as kb_weight<TRON_KB_FAST>.
__device__ __forceinline__ v2f kb_weight_fast2(const v2f x, const KbCoef &k)
{
    const v2f r = x * k.invW;
    const v2f one = {1.0f, 1.0f};
    const v2f s = __builtin_elementwise_fma(-r, r, one);
    v2f acc = {k.poly[0], k.poly[0]};
#pragma unroll
    for (int t = 1; t < kKbPolyTerms; ++t) {
        const v2f c = {k.poly[t], k.poly[t]};
        acc = __builtin_elementwise_fma(acc, s, c);
    }
    if (!(fabsf(x.x) < k.W)) acc.x = 0.0f;
    if (!(fabsf(x.y) < k.W)) acc.y = 0.0f;
    return acc;
}

// streaming (non-temporal) 8-byte load / store: data that is touched once
__device__ __forceinline__ float2 ld_nt(const float2 *p)
{
    const v2f v = __builtin_nontemporal_load(reinterpret_cast<const v2f *>(p));
    return make_float2(v.x, v.y);
}
__device__ __forceinline__ void st_nt(float2 *p, const float2 a)
{
    const v2f v = {a.x, a.y};
    __builtin_nontemporal_store(v, reinterpret_cast<v2f *>(p));
}

// ------------------------------------------------------------------------- LDS-DMA (global -> LDS without registers)

// LDS-only workgroup barrier: every LDS access of this wave has completed; outstanding global loads keep flying.
__device__ __forceinline__ void lds_barrier()
{
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}

__device__ __forceinline__ unsigned lds_addr(const void *q)    // byte address inside the workgroup's LDS allocation
{
    return (unsigned)(size_t)(__attribute__((address_space(3))) const void *)q;
}

// One 16-byte-per-lane copy global -> LDS: lane l's bytes land at lds_dst + 16 l (lds_dst wave-uniform).  Issued from
// inline assembly on purpose: hipcc would otherwise wait vmcnt(0) before the next LDS read of ANY address.
__device__ __forceinline__ void lds_dma16(const void *gsrc, unsigned lds_dst)
{
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(__builtin_amdgcn_readfirstlane((int)lds_dst)) : "memory");
}

// The same with the non-temporal hint (data that is read once, e.g. the FFT intermediate).
__device__ __forceinline__ void lds_dma16_nt(const void *gsrc, unsigned lds_dst)
{
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off nt\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(__builtin_amdgcn_readfirstlane((int)lds_dst)) : "memory");
}

// The same with the source as a wave-uniform base (an SGPR pair) plus a 32-bit byte offset per lane: no 64-bit vector
// address arithmetic at the call site.
__device__ __forceinline__ void lds_dma16_s(const void *sbase, unsigned voff, unsigned lds_dst)
{
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(sbase), "s"(lds_dst) : "memory");
}

// 4 bytes per lane (lane l's bytes land at lds_dst + 4 l): one coil's real or imaginary parts
__device__ __forceinline__ void lds_dma4_s(const void *sbase, unsigned voff, unsigned lds_dst)
{
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dword %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(sbase), "s"(lds_dst) : "memory");
}

__device__ __forceinline__ float safe_rcp(float c)
{
    return fabsf(c) > 1e-12f ? 1.0f / c : copysignf(1e12f, c);
}

template <bool HALF>
__device__ __forceinline__ float2 load_sample(const void *base, size_t idx)
{
    if (HALF) {
        const __half2 h = reinterpret_cast<const __half2 *>(base)[idx];
        return __half22float2(h);
    } else {
        return reinterpret_cast<const float2 *>(base)[idx];
    }
}

// (u nro) / nxos of src/tron.cu:517 (the truncating resample of the readout when nro != nxos) for a radius 0 <= u < nxos / 2, in float
// arithmetic: u nro + 1/2 is exact, and the half keeps the quotient 1 / (2 nxos) away from the integers, more than the product's
// rounding moves it (checked for every u of a plan by arc_resample_exact before a plan takes the arc path).  nro_f = nro, inv_nxos =
// 1.0f / nxos; nro == nxos gives u itself.
__host__ __device__ __forceinline__ float arc_sample_of(float u, float nro_f, float inv_nxos)
{
    return truncf(fmaf(u, nro_f, 0.5f) * inv_nxos);
}

}  // namespace tron
