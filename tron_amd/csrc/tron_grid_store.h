// Output store shared by the gridding kernels of the TRON_KB_FAST path (tron_grid_binned.hip, tron_grid_arc.hip).
#pragma once

#include "tron_device.h"

namespace tron {

// Stores the two horizontally adjacent points (X0, Y), (X0+1, Y) of coil c, slice z: v = (re0, im0, re1, im1).
// ACC: adds to what is stored there (the origin-centred inner tile lands on the four centre tiles' corners).
template <bool ACC = false>
__device__ __forceinline__ void store_point_pair(const GridParams &p, int z, int c, int X0, int Y, const float4 v)
{
    const int n = p.nxos, h = n / 2;
    if (Y + h >= n) return;
    const int row = p.out_shift ? (Y < 0 ? Y + n : Y) : Y + h;     // both fftshifts of src/tron.cu:631 folded in
    const int colA = p.out_shift ? (X0 < 0 ? X0 + n : X0) : X0 + h;
    // one 16-byte store for the two columns needs an even first column; then X0 is even too (n is), so X0 != -1 and
    // the pair does not straddle the periodic wrap.  n/2 odd (e.g. nxos 18, 150) makes every X0 odd: scalar stores.
    const bool pair = (X0 + 1 + h < n) && p.out_p == 1 && (colA & 1) == 0;
    float2 *o = p.udata + (size_t)z * p.out_z + ((size_t)row * n + colA) * p.out_p + (size_t)c * p.out_c;
    if (pair && (n & 1) == 0) {
        float4 w = v;
        if (ACC) {
            const float4 old = *reinterpret_cast<const float4 *>(o);
            w.x += old.x; w.y += old.y; w.z += old.z; w.w += old.w;
        }
        *reinterpret_cast<float4 *>(o) = w;
    } else {
        if (X0 + h < n) o[0] = ACC ? make_float2(o[0].x + v.x, o[0].y + v.y) : make_float2(v.x, v.y);
        if (X0 + 1 + h < n) {
            const int colB = p.out_shift ? (X0 + 1 < 0 ? X0 + 1 + n : X0 + 1) : X0 + 1 + h;
            float2 *ob = p.udata + (size_t)z * p.out_z + ((size_t)row * n + colB) * p.out_p + (size_t)c * p.out_c;
            *ob = ACC ? make_float2(ob->x + v.z, ob->y + v.w) : make_float2(v.z, v.w);
        }
    }
}

}  // namespace tron
