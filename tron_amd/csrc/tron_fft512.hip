// Pruned, fused 2-D inverse FFT for the adjoint tail at the metric size: 512x512 grid -> centre 256x256.
//
// Replaces, for nxos = 512 / nx = 256, the chain  cufftExecC2C(INVERSE) -> fftshift -> crop ->
// deapodkernel -> coilcombinesos  of the reference (src/tron.cu:632-635, 764).  Same unnormalised
// DFT with the +i exponent (CUFFT_INVERSE), same index rotations, same 1/w table and coil order as
// post_kernel, but only what the cropped image needs is computed and moved:
//   pass 1  row FFTs of all 512 rows, keeping the 256 output columns that survive the crop
//           (read 2 MiB, write 1 MiB per coil image), written TRANSPOSED so that
//   pass 2  column FFTs read contiguous 4 KiB lines; only the 256 surviving columns are transformed
//           and only the 256 surviving rows are kept; deapodisation and the root-sum-of-squares over
//           coils happen in registers (read 1 MiB per coil image, write 0.5 MiB per slice).
// 4.06 MB of HBM traffic per coil image instead of the 8.5 MB of a full FFT + separate tail.
//
// A 512-point line is one wave: 64 lanes x 8 points, three radix-8 stages (512 = 8*8*8), two
// exchanges through a private 4.6 KiB LDS region; LDS instructions of one wave execute in order, so
// the line needs no barrier.
#include "tron_device.h"
#include "tron_host.h"

namespace tron {

constexpr int kF = 512;         // line length
constexpr int kFKeep = 256;     // outputs kept per line
constexpr int kLinesPerWg = 16; // lines per workgroup (4 waves x 4 lines): 128-byte transposed segments
constexpr int kPA = 72;         // LDS pitch (float2 units) of the first exchange: stage B's loads fall on 32 distinct bank pairs
constexpr int kXch = 8 * kPA;   // exchange region per wave (float2), >= the 512 points of the second exchange

// One ds_read_b64 at LDS byte address a.  volatile: hipcc pairs neighbouring 8-byte LDS loads into ds_read2_b64 /
// ds_read2st64_b64, which move 128 B per clock on a 32-bank modulus where ds_read_b64 moves 256 on 64 banks -- the FFT
// kernels kept the LDS busy for 73 % of their time, 41 % of that in bank conflicts.
__device__ __forceinline__ float2 lds_ld64(const unsigned a)
{
    const v2f t = *(const volatile __attribute__((address_space(3))) v2f *)(size_t)a;
    return make_float2(t.x, t.y);
}

// Second exchange (stage B -> stage C), point (j1, k1, m2) of the 8 x 8 x 8 cube: the eight points a stage-C thread reads
// lie 32 apart, and within a 32-point block the slot is (4 k1 + (j1 & 3)) ^ g(m2).  Stage C's loads (m2 fixed, lanes over
// k1 and j1 & 3) then fall on 32 distinct bank pairs; stage B's stores (j1 fixed, 16-lane groups over m2 and k1 & 1) on 16.
__device__ __forceinline__ int xch2_index(const int j1, const int k1, const int m2)
{
    return (j1 >> 2) * 256 + m2 * 32 + ((k1 * 4 + (j1 & 3)) ^ ((m2 & 3) | ((m2 >> 2) << 3)));
}

__device__ __forceinline__ float2 cmul(const float2 a, const float2 w)
{
    return make_float2(fmaf(a.x, w.x, -a.y * w.y), fmaf(a.x, w.y, a.y * w.x));
}

__device__ __forceinline__ float2 cadd(const float2 a, const float2 b) { return make_float2(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ float2 csub(const float2 a, const float2 b) { return make_float2(a.x - b.x, a.y - b.y); }
// multiplication by +i (inverse transform)
__device__ __forceinline__ float2 muli(const float2 a) { return make_float2(-a.y, a.x); }

// 8-point DFT with exponent +2*pi*i*n*k/8, natural order in and out
__device__ __forceinline__ void dft8_inv(float2 v[8])
{
    const float h = 0.70710678118654752440f;
    // even / odd 4-point transforms
    const float2 e0 = cadd(v[0], v[4]), e1 = csub(v[0], v[4]), e2 = cadd(v[2], v[6]), e3 = muli(csub(v[2], v[6]));
    const float2 o0 = cadd(v[1], v[5]), o1 = csub(v[1], v[5]), o2 = cadd(v[3], v[7]), o3 = muli(csub(v[3], v[7]));
    const float2 E0 = cadd(e0, e2), E2 = csub(e0, e2), E1 = cadd(e1, e3), E3 = csub(e1, e3);
    const float2 O0 = cadd(o0, o2), O2 = csub(o0, o2), O1 = cadd(o1, o3), O3 = csub(o1, o3);
    // twiddles w8^k, w8 = exp(+i*pi/4)
    const float2 T1 = make_float2(h * (O1.x - O1.y), h * (O1.x + O1.y));       // O1 * (1+i)/sqrt2
    const float2 T2 = muli(O2);                                                // O2 * i
    const float2 T3 = make_float2(-h * (O3.x + O3.y), h * (O3.x - O3.y));      // O3 * (-1+i)/sqrt2
    v[0] = cadd(E0, O0); v[4] = csub(E0, O0);
    v[1] = cadd(E1, T1); v[5] = csub(E1, T1);
    v[2] = cadd(E2, T2); v[6] = csub(E2, T2);
    v[3] = cadd(E3, T3); v[7] = csub(E3, T3);
}

// Orders one wave's LDS exchange: the stores before it are visible to every lane's loads after it.  Lanes of a wave
// exchange data through LDS here with no workgroup barrier; under the HIP memory model that needs a wavefront-scope
// release/acquire pair (no instruction on gfx950 beyond the s_waitcnt the compiler emits anyway: LDS operations of one
// wave execute in order) plus a wave barrier that keeps the compiler from moving accesses across it.
__device__ __forceinline__ void wave_lds_fence()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// The butterflies in packed fp32 (one v_pk_* instruction per complex operation), with the half swaps and sign flips of the
// multiplications by +-i, (1 + i) / sqrt2 and by a twiddle expressed as operand modifiers (op_sel / neg) instead of moves: hipcc
// packs the plain complex adds by itself but spent 52 v_mov and 64 unpacked operations per 512-point line on the rest
// (315 -> 170 VALU instructions per line).  Every result is the same float operation on the same operands as in dft8_inv /
// cmul above: bit-identical output (dft8_inv / cmul stay as the readable statement of what is computed).
__device__ __forceinline__ v2f pk_add_i(const v2f a, const v2f b)          // a + i b = (a.x - b.y, a.y + b.x)
{
    v2f d;
    asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1]" : "=v"(d) : "v"(a), "v"(b));
    return d;
}
__device__ __forceinline__ v2f pk_sub_i(const v2f a, const v2f b)          // a - i b = (a.x + b.y, a.y - b.x)
{
    v2f d;
    asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]" : "=v"(d) : "v"(a), "v"(b));
    return d;
}
__device__ __forceinline__ v2f pk_xmy_xpy(const v2f a)                     // (a.x - a.y, a.x + a.y)
{
    v2f d;
    asm("v_pk_add_f32 %0, %1, %1 op_sel:[0,1] op_sel_hi:[0,1] neg_lo:[0,1]" : "=v"(d) : "v"(a));
    return d;
}
__device__ __forceinline__ v2f pk_xpy_xmy(const v2f a)                     // (a.x + a.y, a.x - a.y)
{
    v2f d;
    asm("v_pk_add_f32 %0, %1, %1 op_sel:[0,1] op_sel_hi:[0,1] neg_hi:[0,1]" : "=v"(d) : "v"(a));
    return d;
}
__device__ __forceinline__ v2f pk_scale(const v2f a, const float h)        // (h a.x, h a.y)
{
    v2f d;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[1,0]" : "=v"(d) : "v"(a), "v"((v2f){h, h}));
    return d;
}
__device__ __forceinline__ v2f pk_scale_nlo(const v2f a, const float h)    // (-h a.x, h a.y)
{
    v2f d;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[1,0] neg_lo:[0,1]" : "=v"(d) : "v"(a), "v"((v2f){h, h}));
    return d;
}
__device__ __forceinline__ v2f pk_cmul(const v2f a, const v2f w)           // (a.x w.x - a.y w.y, a.x w.y + a.y w.x), rounded as cmul
{
    v2f p, d;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel:[1,1] op_sel_hi:[1,0]" : "=v"(p) : "v"(a), "v"(w));                              // (a.y w.y, a.y w.x)
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[0,1,1] neg_lo:[0,0,1]" : "=v"(d) : "v"(a), "v"(w), "v"(p));             // (a.x w.x - p.x, a.x w.y + p.y)
    return d;
}

__device__ __forceinline__ void dft8_inv_pk(v2f v[8])
{
    const float h = 0.70710678118654752440f;
    const v2f e0 = v[0] + v[4], e1 = v[0] - v[4], e2 = v[2] + v[6], d26 = v[2] - v[6];
    const v2f o0 = v[1] + v[5], o1 = v[1] - v[5], o2 = v[3] + v[7], d37 = v[3] - v[7];
    const v2f E0 = e0 + e2, E2 = e0 - e2, E1 = pk_add_i(e1, d26), E3 = pk_sub_i(e1, d26);
    const v2f O0 = o0 + o2, O2 = o0 - o2, O1 = pk_add_i(o1, d37), O3 = pk_sub_i(o1, d37);
    const v2f T1 = pk_scale(pk_xmy_xpy(O1), h);                                // O1 (1 + i) / sqrt2
    const v2f T3 = pk_scale_nlo(pk_xpy_xmy(O3), h);                            // O3 (-1 + i) / sqrt2
    v[0] = E0 + O0; v[4] = E0 - O0;
    v[1] = E1 + T1; v[5] = E1 - T1;
    v[2] = pk_add_i(E2, O2); v[6] = pk_sub_i(E2, O2);
    v[3] = E3 + T3; v[7] = E3 - T3;
}

// 512-point inverse DFT of one line held as v[q] = x[64*q + lane]; returns v[j2] = X[lane + 64*j2].  xch: this wave's private
// LDS exchange region.  twa(k1) = w512^(lane k1), twb(j1) = w512^(8 (lane & 7) j1), w512 = exp(+2 pi i / 512): both depend on
// the lane only, so a kernel that transforms many lines per wave can keep them in registers (fft512_lane_twiddles).
template <class TwA, class TwB>
__device__ __forceinline__ void fft512_inv_tw(float2 vf[8], float2 *xch, const TwA &twa, const TwB &twb, const int lane)
{
    v2f v[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) v[q] = (v2f){vf[q].x, vf[q].y};
    auto st = [&](const int i, const v2f a) { xch[i] = make_float2(a.x, a.y); };
    auto ld = [&](const unsigned a) { const float2 t = lds_ld64(a); return (v2f){t.x, t.y}; };
    // stage A: DFT over n1 (stride 64), twiddle w512^(n2*k1), exchange
    dft8_inv_pk(v);
#pragma unroll
    for (int k1 = 1; k1 < 8; ++k1) v[k1] = pk_cmul(v[k1], twa(k1));
#pragma unroll
    for (int k1 = 0; k1 < 8; ++k1) st(k1 * kPA + lane, v[k1]);
    wave_lds_fence();
    const unsigned xa = lds_addr(xch);
    // stage B: thread (k1 = lane>>3, m2 = lane&7): DFT over m1, twiddle w64^(m2*j1)
    {
        const int k1 = lane >> 3, m2 = lane & 7;
#pragma unroll
        for (int m1 = 0; m1 < 8; ++m1) v[m1] = ld(xa + (unsigned)((k1 * kPA + m1 * 8 + m2) * (int)sizeof(float2)));
        wave_lds_fence();                                                  // stage B's stores reuse the region at another pitch
        dft8_inv_pk(v);
#pragma unroll
        for (int j1 = 1; j1 < 8; ++j1) v[j1] = pk_cmul(v[j1], twb(j1));
#pragma unroll
        for (int j1 = 0; j1 < 8; ++j1) st(xch2_index(j1, k1, m2), v[j1]);  // B[j1][k1*8 + m2]
    }
    wave_lds_fence();
    // stage C: thread (k1 = lane&7, j1 = lane>>3): DFT over m2 -> X[k1 + 8*j1 + 64*j2]
    {
        const int k1 = lane & 7, j1 = lane >> 3;
#pragma unroll
        for (int m2 = 0; m2 < 8; ++m2) v[m2] = ld(xa + (unsigned)(xch2_index(j1, k1, m2) * (int)sizeof(float2)));
        wave_lds_fence();                                                  // the next line's stage A stores come after these loads
        dft8_inv_pk(v);
    }
#pragma unroll
    for (int q = 0; q < 8; ++q) vf[q] = make_float2(v[q].x, v[q].y);
}

// ... with the twiddles read from the table tw[k] = exp(+2*pi*i*k/512) (LDS or global memory) as they are needed
__device__ __forceinline__ void fft512_inv(float2 vf[8], float2 *xch, const float2 *__restrict__ tw, const int lane)
{
    auto twv = [&](const int i) { const float2 t = tw[i & 511]; return (v2f){t.x, t.y}; };
    const int m2 = lane & 7;
    fft512_inv_tw(vf, xch, [&](const int k1) { return twv(lane * k1); }, [&](const int j1) { return twv(8 * m2 * j1); }, lane);
}

// the fourteen twiddles of a lane: twa[k1], twb[j1] for fft512_inv_tw (index 0 unused)
__device__ __forceinline__ void fft512_lane_twiddles(const float2 *__restrict__ tw, const int lane, v2f (&twa)[8], v2f (&twb)[8])
{
    const int m2 = lane & 7;
#pragma unroll
    for (int k = 1; k < 8; ++k) {
        const float2 a = tw[(lane * k) & 511], b = tw[(8 * m2 * k) & 511];
        twa[k] = (v2f){a.x, a.y};
        twb[k] = (v2f){b.x, b.y};
    }
    twa[0] = twb[0] = (v2f){1.f, 0.f};
}

struct Fft512Params {
    const float2 *in;        // pass 1: [img][512][512]; pass 2: [img][256][512] (transposed, pruned)
    float2 *tmp;             // pass 1 output
    float2 *out;             // pass 2 output [slice][256*256]
    const float2 *tw;        // exp(+2 pi i k / 512), k = 0..511
    const float *inv_deapod; // 256*256
    int nchan, nslices;
    int rzero2;              // pass 1: elements with X^2 + Y^2 > rzero2 are known zeros and are not loaded (<= 0: load all)
    // uncombined output (CGNR, Walsh, nt > 1): pass 2 runs per coil image and writes out[slice][nchan * (row * 256 + col) + c],
    // times `scale`; `partial` (may be null): [slice][nchan * column blocks] sums of |out|^2, one per workgroup
    float scale;
    double *partial;
};

// cropped index of kept output k (k < 128 or k >= 384), cf. post_kernel: mr = (row + w - n/2 + n) % n
__device__ __forceinline__ int crop_index(int k) { return k < kFKeep / 2 ? k + kFKeep / 2 : k - (kF - kFKeep / 2); }

// grid = (512/16, nimg); block = 256
// A wave transforms four lines.  A line's loads used to be issued and waited for before its transform (four waves per SIMD
// hid too little of the HBM latency: 3.6 TB/s of the kernel's own traffic, where a plain copy with the same access shape
// reaches 5.5-5.8, tools/probe/copywidth.hip).  Now line j+1 is copied global -> LDS by LDS-DMA while line j is transformed:
// no registers (prefetching into registers cost a wave per SIMD: measured slower), and no LDS either -- the four 4 KiB
// line buffers live in the part of the transposition tile the exchange regions leave unused until the end.  The
// twiddles come from LDS so that no compiler-tracked global load (whose wait would drain the DMA, the counters being
// in-order) sits between the copy's issue and its use.
__global__ void __launch_bounds__(256) fft512_rows_kernel(const Fft512Params p)
{
    __shared__ float2 s_t[kFKeep * (kLinesPerWg + 1)];     // exchange regions | line buffers, then [kept col][line], +1 pad
    __shared__ float2 s_tw[kF];
    static_assert(kFKeep * (kLinesPerWg + 1) >= 4 * kXch + 4 * kF, "exchange regions + line buffers must fit the transposition tile");
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const size_t img = blockIdx.y;
    const int row0 = blockIdx.x * kLinesPerWg;
    const float2 *src = p.in + img * (size_t)kF * kF;
    float2 *xch = s_t + wave * kXch;
    float2 *lbuf = s_t + 4 * kXch + wave * kF;            // this wave's line buffer: the line as it lies in memory
    // the gridded spokes fill a disc of radius nxos/2 - 1 + W (src/tron.cu:498-502): 21 % of the square is zero and is
    // neither copied nor read from the buffer
    auto row_lim = [&](const int row) {
        const int Y = row < kF / 2 ? row : row - kF;
        return p.rzero2 > 0 ? p.rzero2 - Y * Y : 0x7fffffff;
    };
    auto inside = [&](const int col, const int lim) {
        const int X = col < kF / 2 ? col : col - kF;
        return X * X <= lim;
    };
    auto copy_line = [&](const int lr) {                    // 4 x (64 lanes x 16 bytes = two points per lane)
        const int lim = row_lim(row0 + lr);
        const float2 *line = src + (size_t)(row0 + lr) * kF;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int col = k * 128 + 2 * lane;
            if (inside(col, lim) || inside(col + 1, lim)) lds_dma16(line + col, lds_addr(lbuf) + (unsigned)(k * 128 * sizeof(float2)));
        }
    };
    copy_line(wave * 4);
    for (int i = threadIdx.x; i < kF; i += 256) s_tw[i] = p.tw[i];
    __syncthreads();
    float2 keep[4][4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int lr = wave * 4 + j;
        const int lim = row_lim(row0 + lr);
        float2 v[8];
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");    // line j has landed
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const float2 t = lds_ld64(lds_addr(lbuf) + (unsigned)((q * 64 + lane) * (int)sizeof(float2)));
            v[q] = inside(q * 64 + lane, lim) ? t : make_float2(0.f, 0.f);
        }
        if (j < 3) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the buffer has been read: line j+1 may overwrite it
            copy_line(lr + 1);
        }
        fft512_inv(v, xch, s_tw, lane);
        // keep k = lane + 64*j2 for j2 in {0,1,6,7}
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) keep[j][jj] = v[jj < 2 ? jj : jj + 4];
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) {
            const int j2 = jj < 2 ? jj : jj + 4;
            s_t[crop_index(lane + 64 * j2) * (kLinesPerWg + 1) + wave * 4 + j] = keep[j][jj];
        }
    __syncthreads();
    // transposed store: tmp[img][col][row0 .. row0+15]  (128 contiguous bytes per column)
    float2 *dst = p.tmp + img * (size_t)kFKeep * kF;
    for (int e = threadIdx.x; e < kFKeep * kLinesPerWg; e += 256) {
        const int col = e / kLinesPerWg, r = e % kLinesPerWg;
        st_nt(&dst[(size_t)col * kF + row0 + r], s_t[col * (kLinesPerWg + 1) + r]);
    }
}

// grid = (256/16, nslices); block = 256.  Column FFTs + crop + deapodise + root-sum-of-squares.
// SINGLE (one channel): the deapodised complex image passes through (src/tron.cu:265-266).  Otherwise only sum |.|^2 is
// carried across the coils and the (positive) deapodisation factor is applied once at the end, sqrt(sum |v|^2) / w =
// sqrt(sum |v / w|^2) up to fp32 rounding: 16 accumulators instead of 16 + 16 factors + 32 pass-through values per
// thread took the kernel from 182 VGPRs (2 waves per SIMD) to 4 waves per SIMD.
// The next line (4 KiB, contiguous) is copied global -> LDS by LDS-DMA while the current one is transformed, as in pass 1:
// 16 registers fewer than the register prefetch it replaces, so the kernel fits four waves per SIMD.
// LPW = columns per wave (4 LPW per workgroup).  A 64-slice launch has 16 x 64 = 1 024 workgroups of 16 columns = exactly
// the 4 per CU that fit; launches of fewer slices use 8 or 4 columns per workgroup so that the chip is still full (a
// 32-slice launch at 16 columns left half of it idle: 7 % of `bench.py --slices 32`).
// COILS: one coil image per workgroup (grid.x = column blocks x coils), complex output interleaved by coil.  The coils of one
// column block get workgroup ids 8 apart -- the same XCD, close in time -- so that XCD's L2 merges their 8-byte pieces of a
// pixel's nchan * 8 bytes before they reach HBM.
template <bool SINGLE, int LPW, bool COILS = false>
__global__ void __launch_bounds__(256, 3) fft512_cols_post_kernel(const Fft512Params p)
{
    static_assert(!COILS || SINGLE, "uncombined output keeps the complex value");
    constexpr int kCols = 4 * LPW;
    constexpr int kTileElems = kFKeep * (kCols + 1);       // [kept row][col in block], +1 pad
    constexpr int kLdsElems = kTileElems > 4 * kXch + 8 * kF ? kTileElems : 4 * kXch + 8 * kF;
    __shared__ float2 s_t[kLdsElems];                      // exchange regions | two line buffers per wave, then the output tile
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int z = blockIdx.y;
    // COILS: blockIdx.x = (column block & 7) | coil << 3 | (column block >> 3) * 8 nchan
    const int coil = COILS ? (int)((blockIdx.x >> 3) % (unsigned)p.nchan) : 0;
    const int cblk = COILS ? (int)((blockIdx.x & 7u) + 8u * (blockIdx.x / (8u * (unsigned)p.nchan))) : (int)blockIdx.x;
    const int col0 = cblk * kCols;
    const int nloop = COILS ? 1 : p.nchan;
    float2 *xch = s_t + wave * kXch;
    const unsigned lbuf = lds_addr(s_t + 4 * kXch + wave * 2 * kF);           // this wave's two line buffers
    float val[LPW][4];
    float2 single[SINGLE ? LPW : 1][4];
#pragma unroll
    for (int j = 0; j < LPW; ++j)
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) {
            val[j][jj] = 0.f;
            if (SINGLE) single[j][jj] = make_float2(0.f, 0.f);
        }
    // The pass is bound by memory latency, not arithmetic (packed butterflies: -27 % instructions, -1 % time): every wave keeps TWO
    // lines (4 KiB each, contiguous) on their way global -> LDS by LDS-DMA while it transforms a third.  The LDS for the second
    // buffer comes from the twiddle table, whose fourteen entries per lane wait in registers instead (no load between a copy's
    // issue and its use: the counters are in-order), and from running three workgroups per CU, not four.
    v2f twa[8], twb[8];
    fft512_lane_twiddles(p.tw, lane, twa, twb);
    const float2 *base = p.in + ((size_t)z * p.nchan + coil) * (size_t)kFKeep * kF + (size_t)(col0 + wave * LPW) * kF;
    const int nlines = nloop * LPW;                         // line i = (coil i / LPW, column i % LPW) of this wave
    auto copy_line = [&](const int i) {
        // the intermediate is written once by pass 1 and read once here: streaming (non-temporal) accesses on both sides
        const float2 *line = base + (size_t)(i / LPW) * kFKeep * kF + (size_t)(i % LPW) * kF;
        const unsigned dst = lbuf + (unsigned)((i & 1) * kF * (int)sizeof(float2));
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            lds_dma16_nt(line + k * 128 + 2 * lane, dst + (unsigned)(k * 128 * sizeof(float2)));
        }
    };
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // the twiddles are in: from here on only the copies count
    copy_line(0);
    if (nlines > 1) copy_line(1);
    for (int c = 0; c < nloop; ++c) {
#pragma unroll
        for (int j = 0; j < LPW; ++j) {
            const int i = c * LPW + j;
            float2 v[8];
            if (i + 1 < nlines) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");   // line i has landed (the four pieces of line i + 1 may still fly)
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            const unsigned src = lbuf + (unsigned)((i & 1) * kF * (int)sizeof(float2));
#pragma unroll
            for (int q = 0; q < 8; ++q) v[q] = lds_ld64(src + (unsigned)((q * 64 + lane) * (int)sizeof(float2)));
            if (i + 2 < nlines) {
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");       // the buffer has been read
                copy_line(i + 2);
            }
            fft512_inv_tw(v, xch, [&](const int k) { return twa[k]; }, [&](const int k) { return twb[k]; }, lane);
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) {
                const int j2 = jj < 2 ? jj : jj + 4;
                const float2 u = v[j2];
                if (SINGLE) single[j][jj] = u;
                else val[j][jj] += u.x * u.x + u.y * u.y;                     // src/tron.cu:262 (the factor 1/w^2 is applied below)
            }
        }
    }
    __syncthreads();                                       // every wave is done with its exchange region and line buffer
    double nrm = 0.0;
#pragma unroll
    for (int j = 0; j < LPW; ++j)
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) {
            const int j2 = jj < 2 ? jj : jj + 4;
            const int rowc = crop_index(lane + 64 * j2);
            float inv = p.inv_deapod[rowc * kFKeep + col0 + wave * LPW + j];         // src/tron.cu:398-400
            if (COILS) inv *= p.scale;
            const float2 o = SINGLE ? make_float2(single[j][jj].x * inv, single[j][jj].y * inv)     // src/tron.cu:259-266
                                    : make_float2(sqrtf(val[j][jj]) * inv, 0.f);
            if (COILS) nrm += (double)o.x * o.x + (double)o.y * o.y;
            s_t[rowc * (kCols + 1) + wave * LPW + j] = o;
        }
    __syncthreads();
    if (COILS) {
        float2 *dst = p.out + (size_t)z * kFKeep * kFKeep * p.nchan + coil;
        for (int e = threadIdx.x; e < kFKeep * kCols; e += 256) {
            const int row = e / kCols, cc = e % kCols;
            dst[((size_t)row * kFKeep + col0 + cc) * p.nchan] = s_t[row * (kCols + 1) + cc];
        }
        if (p.partial) {                                   // |out|^2 of this workgroup's share, summed in a fixed order
            __syncthreads();
            double *sm = reinterpret_cast<double *>(s_t);
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) nrm += __shfl_down(nrm, o);
            if (lane == 0) sm[wave] = nrm;
            __syncthreads();
            if (threadIdx.x == 0)
                p.partial[((size_t)z * p.nchan + coil) * (kFKeep / kCols) + cblk] = ((sm[0] + sm[1]) + sm[2]) + sm[3];
        }
        return;
    }
    float2 *dst = p.out + (size_t)z * kFKeep * kFKeep;
    for (int e = threadIdx.x; e < kFKeep * kCols; e += 256) {
        const int row = e / kCols, cc = e % kCols;
        dst[(size_t)row * kFKeep + col0 + cc] = s_t[row * (kCols + 1) + cc];
    }
}

hipError_t launch_fft512_adjoint(const float2 *grid, float2 *tmp, float2 *out, const float2 *tw, const float *inv_deapod, int rzero,
                                 int nchan, int nslices, hipStream_t s)
{
    Fft512Params p;
    p.rzero2 = rzero > 0 ? rzero * rzero : 0;
    p.in = grid; p.tmp = tmp; p.out = out; p.tw = tw; p.inv_deapod = inv_deapod; p.nchan = nchan; p.nslices = nslices;
    p.scale = 1.f; p.partial = nullptr;
    hipLaunchKernelGGL(fft512_rows_kernel, dim3(kF / kLinesPerWg, nslices * nchan), dim3(256), 0, s, p);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    p.in = tmp;
    // 8 columns per workgroup (4 from launches of fewer than 32 slices on, so that the chip is still full): three workgroups per CU hold
    // 768 at a time, and 16 columns each made a 128-slice launch 2.67 rounds of them (221 us; 8 columns: 196, 4 columns: 224)
    int lpw = nslices >= 32 ? 2 : 1;
    if (nchan == 1) lpw = nslices >= 64 ? 4 : lpw;          // one channel: a wave's lines are its columns only, 16 per workgroup amortise its set-up
    const dim3 grid_c(kFKeep / (4 * lpw), nslices);
    if (nchan == 1) {
        if (lpw == 4) hipLaunchKernelGGL((fft512_cols_post_kernel<true, 4>), grid_c, dim3(256), 0, s, p);
        else if (lpw == 2) hipLaunchKernelGGL((fft512_cols_post_kernel<true, 2>), grid_c, dim3(256), 0, s, p);
        else hipLaunchKernelGGL((fft512_cols_post_kernel<true, 1>), grid_c, dim3(256), 0, s, p);
    } else {
        if (lpw == 4) hipLaunchKernelGGL((fft512_cols_post_kernel<false, 4>), grid_c, dim3(256), 0, s, p);
        else if (lpw == 2) hipLaunchKernelGGL((fft512_cols_post_kernel<false, 2>), grid_c, dim3(256), 0, s, p);
        else hipLaunchKernelGGL((fft512_cols_post_kernel<false, 1>), grid_c, dim3(256), 0, s, p);
    }
    return hipGetLastError();
}

// Uncombined variant: out[slice][nchan * (row * 256 + col) + c] = deapodised coil images times `scale`; partial (may be null)
// receives fft512_coils_partials(nslices) sums of |out|^2 per slice, one per workgroup (CGNR's |ztilde|^2, src/tron.cu:695).
int fft512_coils_partials(int nslices)
{
    const int lpw = nslices >= 64 ? 4 : (nslices >= 32 ? 2 : 1);
    return kFKeep / (4 * lpw);                             // per coil
}

hipError_t launch_fft512_adjoint_coils(const float2 *grid, float2 *tmp, float2 *out, const float2 *tw, const float *inv_deapod, int rzero,
                                       int nchan, int nslices, float scale, double *partial, hipStream_t s)
{
    Fft512Params p;
    p.rzero2 = rzero > 0 ? rzero * rzero : 0;
    p.in = grid; p.tmp = tmp; p.out = out; p.tw = tw; p.inv_deapod = inv_deapod; p.nchan = nchan; p.nslices = nslices;
    p.scale = scale; p.partial = partial;
    hipLaunchKernelGGL(fft512_rows_kernel, dim3(kF / kLinesPerWg, nslices * nchan), dim3(256), 0, s, p);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    p.in = tmp;
    const int lpw = nslices >= 64 ? 4 : (nslices >= 32 ? 2 : 1);
    const dim3 grid_c((kFKeep / (4 * lpw)) * nchan, nslices);       // column blocks (a multiple of 8) x coils
    if (lpw == 4) hipLaunchKernelGGL((fft512_cols_post_kernel<true, 4, true>), grid_c, dim3(256), 0, s, p);
    else if (lpw == 2) hipLaunchKernelGGL((fft512_cols_post_kernel<true, 2, true>), grid_c, dim3(256), 0, s, p);
    else hipLaunchKernelGGL((fft512_cols_post_kernel<true, 1, true>), grid_c, dim3(256), 0, s, p);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------ forward head
//
// Pruned, fused forward FFT for nx = 256 / nxos = 512: replaces  pad -> deapodkernel(sigma = 1) ->
// fftshift(FORWARD) -> cufftExecC2C(FORWARD)  of the reference (src/tron.cu:642-645).  The padded image
// is zero outside its centre 255x255 block (pad drops image row/col 0, src/tron.cu:449-450), so
//   pass 1  transforms only the 256 rows that can be non-zero, straight from the coil-interleaved image
//           (read 0.5 MiB, write 1 MiB per coil image);
//   pass 2  transforms all 512 columns of those 256 rows (read 1 MiB, write 2 MiB), 16 columns per
//           workgroup, read in 128-byte segments through an LDS transposition and written as whole lines:
//           the result is stored TRANSPOSED (out[k2][k1]); the degridding kernel swaps its indices.
// 4.5 MiB of HBM traffic per coil image instead of pre_kernel's 2.5 + a full FFT's 8.
// Same unnormalised DFT with the -i exponent as CUFFT_FORWARD, computed as conj(IDFT(conj(x))).

__device__ __forceinline__ float2 cconj(const float2 a) { return make_float2(a.x, -a.y); }

struct Fft512FwdParams {
    const float2 *img;        // [image][nchan*(row*256+col) + c]
    float2 *tmp;              // [image*nchan + c][256 rows][512]
    float2 *out;              // [image*nchan + c][512 k2][512 k1]: transpose of the FFT-native order
    const float2 *tw;
    const float *inv_deapod;  // 512*512, indexed by padded position (src/tron.cu:398-400)
    int nchan;
    int rzero2;               // pass 2: grid points with X^2 + Y^2 > rzero2 (centred) are not stored: no sample's footprint reaches them
                              // (<= 0: store everything)
    int rot;                  // pass 2: point k1 of an output line is stored at (k1 + rot) mod 512 (DegridParams::in_rot)
};

// grid = (256/16, nimg*nchan); block = 256.  Line r of a coil image = padded row 128 + r.
__global__ void __launch_bounds__(256) fft512_fwd_rows_kernel(const Fft512FwdParams p)
{
    __shared__ float2 s_x[4 * kXch];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int ci = blockIdx.y;                                  // image * nchan + coil
    const int k = ci / p.nchan, c = ci - k * p.nchan;
    const float2 *src = p.img + (size_t)k * p.nchan * kFKeep * kFKeep + c;
    float2 *dst = p.tmp + (size_t)ci * kFKeep * kF;
    float2 *xch = s_x + wave * kXch;
    for (int j = 0; j < 4; ++j) {
        const int r = blockIdx.x * kLinesPerWg + wave * 4 + j;  // image row; padded row xdst = r + 128
        float2 v[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) v[q] = make_float2(0.f, 0.f);
        // FFT-input column sc = 64q + lane holds padded column ydst = (sc + 256) % 512 (fftshift, src/tron.cu:164-172);
        // image column y = ydst - 128: q = 0,1 -> y = 128 + sc; q = 6,7 -> y = sc - 384; everything else is padding
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int q = t < 2 ? t : t + 4;
            const int y = t < 2 ? 128 + 64 * q + lane : 64 * (q - 6) + lane;
            if (r > 0 && y > 0) {                                                   // src/tron.cu:449-450
                float2 u = src[((size_t)r * kFKeep + y) * p.nchan];
                const float inv = p.inv_deapod[(size_t)(r + 128) * kF + (y + 128)];
                u.x *= inv; u.y *= inv;
                v[q] = cconj(u);
            }
        }
        fft512_inv(v, xch, p.tw, lane);
        float2 *line = dst + (size_t)r * kF;
#pragma unroll
        for (int j2 = 0; j2 < 8; ++j2) line[lane + 64 * j2] = cconj(v[j2]);
    }
}

// Same pass with coalesced input: a workgroup takes `rows` consecutive image rows with ALL their coils (contiguous in the
// coil-interleaved image), copies them through LDS into per-(row, coil) lines and transforms lines wave by wave.
// grid = (256/rows, nimg); block = 256; dynamic LDS = rows*nchan*kInPitch float2.
constexpr int kInPitch = kFKeep + 4;    // 260: the 8 coils of a pixel land 8 banks apart, consecutive pixels 2 apart

__global__ void __launch_bounds__(256) fft512_fwd_rows_lds_kernel(const Fft512FwdParams p, const int rows)
{
    __shared__ float2 s_x[4 * kXch];
    extern __shared__ __align__(16) float2 s_in[];              // [row in block][coil][kInPitch]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int k = blockIdx.y;
    const int r0 = blockIdx.x * rows;
    const int nlines = rows * p.nchan;
    const int per_row = kFKeep * p.nchan;
    const float2 *src = p.img + ((size_t)k * kFKeep + r0) * per_row;
    for (int e0 = 0; e0 < rows * per_row; e0 += 4 * 256) {      // four coalesced loads in flight per thread
        float2 ld[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int e = e0 + u * 256 + threadIdx.x;
            ld[u] = e < rows * per_row ? src[e] : make_float2(0.f, 0.f);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int e = e0 + u * 256 + threadIdx.x;
            if (e < rows * per_row) {
                const int rl = e / per_row, rem = e - rl * per_row;
                const int y = rem / p.nchan, c = rem - y * p.nchan;
                s_in[(rl * p.nchan + c) * kInPitch + y] = ld[u];
            }
        }
    }
    // the 1/w factors depend on (row, column) only: fetched once per image row and wave, the first set while the
    // image rows are still on their way into LDS
    float inv[4];
    int inv_rl = -1;
    auto load_inv = [&](int rl) {
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int q = t < 2 ? t : t + 4;
            const int y = t < 2 ? 128 + 64 * q + lane : 64 * (q - 6) + lane;
            inv[t] = p.inv_deapod[(size_t)(r0 + rl + 128) * kF + (y + 128)];
        }
        inv_rl = rl;
    };
    if (wave < nlines) load_inv(wave / p.nchan);
    __syncthreads();
    float2 *xch = s_x + wave * kXch;
    for (int L = wave; L < nlines; L += 4) {
        const int rl = L / p.nchan, c = L - rl * p.nchan;
        const int r = r0 + rl;
        const float2 *lin = s_in + L * kInPitch;
        if (rl != inv_rl) load_inv(rl);
        float2 v[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) v[q] = make_float2(0.f, 0.f);
#pragma unroll
        for (int t = 0; t < 4; ++t) {                           // same index map as fft512_fwd_rows_kernel
            const int q = t < 2 ? t : t + 4;
            const int y = t < 2 ? 128 + 64 * q + lane : 64 * (q - 6) + lane;
            if (r > 0 && y > 0) {                                                   // src/tron.cu:449-450
                float2 u = lin[y];
                u.x *= inv[t]; u.y *= inv[t];
                v[q] = cconj(u);
            }
        }
        fft512_inv(v, xch, p.tw, lane);
        float2 *line = p.tmp + ((size_t)(k * p.nchan + c) * kFKeep + r) * kF;
#pragma unroll
        for (int j2 = 0; j2 < 8; ++j2) line[lane + 64 * j2] = cconj(v[j2]);
    }
}

// Pass 1 the same way (see fft512_fwd_cols_dma_kernel below for the scheme): one workgroup of 1 024 threads walks
// kFwdRowSteps steps of `rows` consecutive image rows with all their coils -- a contiguous piece of the coil-interleaved
// image, copied as it lies -- and the 1/w factors of those rows; wave = line (row, coil).  grid = (256 / (rows * steps), nimg).
constexpr int kFwdRowThreads = 1024;
constexpr int kFwdRowSteps = 8;

__global__ void __launch_bounds__(kFwdRowThreads) fft512_fwd_rows_dma_kernel(const Fft512FwdParams p, const int rows)
{
    extern __shared__ __align__(16) float2 s_dyn[];            // exchange regions | twiddles | two image buffers | two 1/w buffers
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float2 *xch = s_dyn + wave * kXch;
    float2 *s_tw = s_dyn + (kFwdRowThreads / 64) * kXch;
    float2 *s_in = s_tw + kF;
    const int belems = rows * kFKeep * p.nchan;                 // float2 per image buffer
    float *s_inv = reinterpret_cast<float *>(s_in + 2 * belems);   // [buf][row in step][256]
    const int k = blockIdx.y;
    const int r0 = blockIdx.x * rows * kFwdRowSteps;
    const int nlines = rows * p.nchan;
    auto fetch = [&](const int step, const int buf) {
        const float2 *src = p.img + ((size_t)k * kFKeep + r0 + step * rows) * kFKeep * p.nchan;
        for (int q0 = wave * 64; q0 < belems / 2; q0 += kFwdRowThreads)
            if (q0 + lane < belems / 2)
                lds_dma16(src + 2 * (q0 + lane), lds_addr(s_in) + (unsigned)((buf * belems + 2 * q0) * sizeof(float2)));
        if (wave < rows)                                        // the row's 1/w factors at padded columns 128 .. 383 (src/tron.cu:398-400)
            lds_dma16(p.inv_deapod + (size_t)(r0 + step * rows + wave + 128) * kF + 128 + 4 * lane,
                      lds_addr(s_inv) + (unsigned)((buf * rows + wave) * kFKeep * sizeof(float)));
    };
    fetch(0, 0);
    for (int i = threadIdx.x; i < kF; i += kFwdRowThreads) s_tw[i] = p.tw[i];
    for (int step = 0; step < kFwdRowSteps; ++step) {
        const int buf = step & 1;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // this thread's pieces of the step have landed
        __syncthreads();                                        // ... everybody's; and nobody still reads the other buffer
        if (step + 1 < kFwdRowSteps) fetch(step + 1, buf ^ 1);
        if (wave < nlines) {
            const int rl = wave / p.nchan, c = wave - rl * p.nchan;
            const int r = r0 + step * rows + rl;
            const unsigned lin = lds_addr(s_in) + (unsigned)((buf * belems + rl * kFKeep * p.nchan + c) * sizeof(float2));
            const unsigned inv0 = lds_addr(s_inv) + (unsigned)((buf * rows + rl) * kFKeep * sizeof(float));
            float2 v[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) v[q] = make_float2(0.f, 0.f);
#pragma unroll
            for (int t = 0; t < 4; ++t) {                       // same index map as fft512_fwd_rows_kernel
                const int q = t < 2 ? t : t + 4;
                const int y = t < 2 ? 128 + 64 * q + lane : 64 * (q - 6) + lane;
                if (r > 0 && y > 0) {                                                   // src/tron.cu:449-450
                    float2 u = lds_ld64(lin + (unsigned)(y * p.nchan * (int)sizeof(float2)));
                    const float inv = *(const volatile __attribute__((address_space(3))) float *)(size_t)(inv0 + (unsigned)(y * 4));
                    u.x *= inv; u.y *= inv;
                    v[q] = cconj(u);
                }
            }
            fft512_inv(v, xch, s_tw, lane);
            float2 *line = p.tmp + ((size_t)(k * p.nchan + c) * kFKeep + r) * kF;
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");    // the next step's copy (issued a transform ago), not the stores below
#pragma unroll
            for (int j2 = 0; j2 < 8; ++j2) line[lane + 64 * j2] = cconj(v[j2]);
        }
    }
}

// Pass 2: column k2 of the row-transformed block, written as one contiguous line -- out is the TRANSPOSE of the FFT-native grid,
// out[k2][k1] (the degridding kernel swaps its indices instead) -- with the next block on its way while the current one is
// transformed.  (Until round 6 a plain form stood beside it that loaded a 256 x 16 block, waited, transformed and stored, four
// workgroups per CU overlapping by chance: 3.5 TB/s of its own traffic, where this one reaches 4.4.)  One workgroup of 1 024 threads (16 waves = the 16 columns
// of a block, four waves per SIMD as there) owns the CU and walks kFwdColBlocks consecutive column blocks of one coil
// image: block b + 1 is copied global -> LDS (global_load_lds_dwordx4: no registers) into the second of two 32 KiB
// buffers while block b is transformed.  A buffer holds the block as it lies in memory, [row][16 columns], the 16-byte
// column pairs of a row XOR-swizzled with the row so that a column read (64 rows, one per lane) is 2-way instead of
// 16-way conflicted.  The copy is waited for BEFORE the block's lines are stored: it has had the transform's time to
// land, and the stores are never waited for.  Twiddles come from LDS (a compiler-tracked global load between the
// copy's issue and its use would make the compiler's s_waitcnt drain the copy).
constexpr int kFwdColThreads = 1024;
constexpr int kFwdColBlocks = 8;                   // column blocks per workgroup
constexpr int kFwdColBuf = kFKeep * kLinesPerWg;   // float2 per buffer
constexpr size_t kFwdColLds = (2 * kFwdColBuf + (kFwdColThreads / 64) * kXch + kF) * sizeof(float2);

__global__ void __launch_bounds__(kFwdColThreads) fft512_fwd_cols_dma_kernel(const Fft512FwdParams p)
{
    extern __shared__ __align__(16) float2 s_dyn[];            // two block buffers | exchange regions | twiddles
    float2 *s_in = s_dyn;
    float2 *xch = s_dyn + 2 * kFwdColBuf + (threadIdx.x >> 6) * kXch;
    float2 *s_tw = s_dyn + 2 * kFwdColBuf + (kFwdColThreads / 64) * kXch;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int ci = blockIdx.y;
    const int b0 = blockIdx.x * kFwdColBlocks;
    const float2 *src = p.tmp + (size_t)ci * kFKeep * kF;
    float2 *dst = p.out + (size_t)ci * kF * kF;
    // piece q = 16 bytes = columns 2 cp, 2 cp + 1 of row q >> 3; it lands at slot (q & 7) of its row, cp = slot ^ ((row >> 1) & 7)
    auto fetch = [&](const int blk, const int buf) {
#pragma unroll
        for (int it = 0; it < kFwdColBuf / 2 / kFwdColThreads; ++it) {
            const int q = it * kFwdColThreads + threadIdx.x;
            const int row = q >> 3, cp = (q & 7) ^ ((row >> 1) & 7);
            lds_dma16_nt(src + (size_t)row * kF + blk * kLinesPerWg + 2 * cp,
                         lds_addr(s_in) + (unsigned)((buf * kFwdColBuf + 2 * (it * kFwdColThreads + wave * 64)) * sizeof(float2)));
        }
    };
    fetch(b0, 0);
    for (int i = threadIdx.x; i < kF; i += kFwdColThreads) s_tw[i] = p.tw[i];
    for (int b = 0; b < kFwdColBlocks; ++b) {
        const int buf = b & 1;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // this thread's pieces of block b have landed
        __syncthreads();                                        // ... everybody's; and nobody still reads the other buffer
        if (b + 1 < kFwdColBlocks) fetch(b0 + b + 1, buf ^ 1);
        // FFT-input row sr = 64q + lane holds padded row (sr + 256) % 512 = 128 + r:  q = 0,1 -> r = 128 + sr; q = 6,7 -> r = sr - 384
        const unsigned in0 = lds_addr(s_in) + (unsigned)(buf * kFwdColBuf * sizeof(float2));
        auto at = [&](const int row) {
            const int cp = (wave >> 1) ^ ((row >> 1) & 7);
            return lds_ld64(in0 + (unsigned)((row * kLinesPerWg + 2 * cp + (wave & 1)) * (int)sizeof(float2)));
        };
        float2 v[8];
        v[0] = cconj(at(128 + lane)); v[1] = cconj(at(192 + lane)); v[6] = cconj(at(lane)); v[7] = cconj(at(64 + lane));
        v[2] = v[3] = v[4] = v[5] = make_float2(0.f, 0.f);
        fft512_inv(v, xch, s_tw, lane);
        const int k2 = (b0 + b) * kLinesPerWg + wave;
        float2 *line = dst + (size_t)k2 * kF;
        const int yc = k2 < kF / 2 ? k2 : k2 - kF;              // FFT-native index -> centred coordinate
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // the copy of block b + 1 (issued a transform ago), not the stores below
#pragma unroll
        for (int j2 = 0; j2 < 8; ++j2) {
            const int k1 = lane + 64 * j2;
            const int xc = k1 < kF / 2 ? k1 : k1 - kF;
            if (p.rzero2 <= 0 || xc * xc + yc * yc <= p.rzero2) line[(k1 + p.rot) & (kF - 1)] = cconj(v[j2]);
        }
    }
}

hipError_t launch_fft512_forward(const float2 *img, float2 *tmp, float2 *out, const float2 *tw, const float *inv_deapod, int rzero, int rot,
                                 int nchan, int nimg, hipStream_t s)
{
    Fft512FwdParams p;
    p.rot = rot;
    p.img = img; p.tmp = tmp; p.out = out; p.tw = tw; p.inv_deapod = inv_deapod; p.nchan = nchan;
    p.rzero2 = rzero > 0 ? rzero * rzero : 0;
    if (nchan > 1 && nchan <= 16) {
        int rows = 16 / nchan;                                  // about 16 lines per workgroup
        if (rows < 1) rows = 1;
        while (kFKeep % rows) --rows;
        if (kFKeep % (rows * kFwdRowSteps) == 0) {
            const size_t lds = ((kFwdRowThreads / 64) * kXch + kF + 2 * (size_t)rows * kFKeep * nchan) * sizeof(float2)
                               + 2 * (size_t)rows * kFKeep * sizeof(float);             // <= 146 KiB
            const hipError_t once = allow_dynamic_lds(reinterpret_cast<const void *>(fft512_fwd_rows_dma_kernel), 160 * 1024);
            if (once != hipSuccess) return once;
            hipLaunchKernelGGL(fft512_fwd_rows_dma_kernel, dim3(kFKeep / (rows * kFwdRowSteps), nimg), dim3(kFwdRowThreads), lds, s, p, rows);
        } else {
            const size_t lds = (size_t)rows * nchan * kInPitch * sizeof(float2);   // <= 33 KiB
            hipLaunchKernelGGL(fft512_fwd_rows_lds_kernel, dim3(kFKeep / rows, nimg), dim3(256), lds, s, p, rows);
        }
    } else {
        hipLaunchKernelGGL(fft512_fwd_rows_kernel, dim3(kFKeep / kLinesPerWg, nimg * nchan), dim3(256), 0, s, p);
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    {
        const hipError_t once = allow_dynamic_lds(reinterpret_cast<const void *>(fft512_fwd_cols_dma_kernel), (int)kFwdColLds);
        if (once != hipSuccess) return once;
        hipLaunchKernelGGL(fft512_fwd_cols_dma_kernel, dim3(kF / kLinesPerWg / kFwdColBlocks, nimg * nchan), dim3(kFwdColThreads), kFwdColLds, s, p);
    }
    return hipGetLastError();
}

__global__ void warm_fft512_tu() {}

hipError_t warm_fft512()   // see warm_kernels() in tron_kernels.hip
{
    hipLaunchKernelGGL(warm_fft512_tu, dim3(1), dim3(64), 0, nullptr);
    return hipGetLastError();
}

}  // namespace tron
